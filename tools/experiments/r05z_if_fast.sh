# the profiles of record again, but only on a box whose probe reads >= 2040 issued TFLOP/s (the pool's boxes differ by 7 %: `box` in the bench line)
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r05z_boxcheck.txt 2>&1
import torch, bench
print(bench.box_probe(torch.device("cuda:0"))["tflops_issued"])
PY
tf=$(tail -1 gpurun_out/r05z_boxcheck.txt)
echo "box probe: $tf TF"
if python -c "import sys; sys.exit(0 if float('$tf') >= 2040 else 1)"; then
  bash tools/experiments/r05z_profiles.sh > gpurun_out/r05z_profiles.log 2>&1
  tail -3 gpurun_out/r05z_profiles.log | cut -c1-200
  echo "PROFILED"
else
  echo "slow box: skipped"
fi
