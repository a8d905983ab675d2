# usage: bash tools/experiments/full_profile.sh <tag>: bench line + kernel-trace stats + FETCH/WRITE PMC passes + SQ PMC pass, all into gpurun_out/
tag=${1:-x}
R=$GRAFT_REPO_ROOT
python bench.py > $R/gpurun_out/${tag}_bench.log 2>&1
tail -1 $R/gpurun_out/${tag}_bench.log > $R/gpurun_out/${tag}_bench.json
cut -c1-400 $R/gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${tag}_mfma -- python3 $R/tools/kbench.py --iters 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/prof_${tag}_lds -- python3 $R/tools/kbench.py --iters 1 > /dev/null 2>&1
find $R/gpurun_out -path "*prof_${tag}_*" -name "*.csv" | head -20
