# round 4, first probe: where the time of single-utterance synthesis goes (kernel durations vs gaps), WaveGlow and WaveFlow
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/experiments/infer_latency.py 63 > $R/gpurun_out/r04a_infer_latency.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04a_infer -- python3 $R/tools/experiments/infer_profile.py 63 > $R/gpurun_out/r04a_infer.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04a_infer 0.5 30 > $R/gpurun_out/r04a_infer_gaps.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04a_infer10 -- python3 $R/tools/experiments/infer_profile.py 862 > $R/gpurun_out/r04a_infer10.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04a_infer10 0.5 30 > $R/gpurun_out/r04a_infer10_gaps.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04a_wfinv -- python3 $R/tools/experiments/wf_infer_profile.py 16128 1 > $R/gpurun_out/r04a_wfinv.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04a_wfinv 0.5 30 > $R/gpurun_out/r04a_wfinv_gaps.txt 2>&1
rm -rf $R/gpurun_out/prof_r04a_wfinv $R/gpurun_out/prof_r04a_infer10
cat $R/gpurun_out/r04a_infer_gaps.txt $R/gpurun_out/r04a_infer10_gaps.txt $R/gpurun_out/r04a_wfinv_gaps.txt
tail -3 $R/gpurun_out/r04a_infer.log $R/gpurun_out/r04a_infer10.log $R/gpurun_out/r04a_wfinv.log
head -5 $R/gpurun_out/r04a_infer_latency.log
