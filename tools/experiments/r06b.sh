# round 6: kernel statistics of the 10 s synthesis call (862 frames -> 220 672 samples) and of the musicnet architecture; GPU tests of the new pieces
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06b_inf -- python3 $R/tools/experiments/infer_profile.py 862 > $R/gpurun_out/r06b_inf.log 2>&1
st=$(ls -t $(find $R/gpurun_out/prof_r06b_inf -name "*kernel_stats.csv") | head -1)
cp $st $R/gpurun_out/r06b_inf_kernel_stats.csv
rm -rf $R/gpurun_out/prof_r06b_inf
cd $R
tail -2 gpurun_out/r06b_inf.log
head -14 gpurun_out/r06b_inf_kernel_stats.csv | cut -c1-200
python -m pytest tests/test_gpu_parity.py -x -q -k "any_transform or wsrglow_timed or shape_sweep or test_one_launch_layer or layer_as_one_launch or inverse_seam or cut_along_k" 2>&1 | tail -5 > gpurun_out/r06b_tests.txt
cat gpurun_out/r06b_tests.txt
