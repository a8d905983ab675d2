# usage: bash tools/experiments/full_profile_model.sh <tag> <model>: `bench.py --model <model>` line + kernel-trace stats + FETCH / WRITE PMC
# passes + SQ PMC passes of the same command, all into gpurun_out/ (summaries: tools/profile_summary.py)
tag=${1:-x}; model=${2:-waveflow}
R=$GRAFT_REPO_ROOT
python $R/bench.py --model $model > $R/gpurun_out/${tag}_bench.log 2>&1
tail -1 $R/gpurun_out/${tag}_bench.log > $R/gpurun_out/${tag}_bench.json
cut -c1-300 $R/gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -- python3 $R/bench.py --model $model --no-inverse --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py --model $model --no-inverse --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -- python3 $R/bench.py --model $model --no-inverse --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${tag}_mfma -- python3 $R/bench.py --model $model --no-inverse --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/prof_${tag}_lds -- python3 $R/bench.py --model $model --no-inverse --steps 1 --warmup 1 > /dev/null 2>&1
cd $R
st=$(ls -t $(find gpurun_out/prof_${tag}_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_${tag}_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_${tag}_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py $tag $st $fe $wr
python tools/profile_summary.py --pmc $tag $(find gpurun_out/prof_${tag}_mfma gpurun_out/prof_${tag}_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes $tag gpurun_out/${tag}_bench.json || true
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/ && cp gpurun_out/${tag}_bench.json gpurun_out/profiles_$tag/
rm -rf gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write gpurun_out/prof_${tag}_mfma gpurun_out/prof_${tag}_lds
head -14 profiles/${tag}_kernel_stats.csv
