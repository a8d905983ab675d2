# samples rocm-smi power / clocks WHILE the training step runs (bench.py with many steps in the background)
python bench.py --steps 400 --warmup 3 --no-cpu --no-extra --no-inverse > /tmp/bench_out.txt 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|Sensor junction" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 1; done
wait $BP
tail -1 /tmp/bench_out.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
