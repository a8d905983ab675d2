# samples rocm-smi power / clocks while the training step (or $1) runs
(WGFLOW_LIB=${WGFLOW_LIB:-} python bench.py --steps 60 --warmup 3 --no-cpu --no-extra --no-inverse > /tmp/bench_out.txt 2>/dev/null &) 
sleep 20
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 1; done
wait
tail -1 /tmp/bench_out.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i power
