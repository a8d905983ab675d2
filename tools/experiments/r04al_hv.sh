cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "waveflow or wn2d" > gpurun_out/r04al_test.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04al_test.txt
python bench.py --model waveflow --no-inverse --steps 8 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r04al_wf.json
python -c "
import json; d=json.load(open('gpurun_out/r04al_wf.json')); print('waveflow', round(d['ms_per_step'],2), [(k['kernel_class'][:14],k['M'],k['K'],round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:6]])"
