"""prints the per-shape table of a bench.py JSON line read from stdin (developer tool)"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.2f ms/step  gate %.1f us  loss %.6f' % (d['ms_per_step'], d['roofline']['launch_ms'] * 1e3, d['loss']))
for k in d['roofline']['kernels']['kernels']:
    print('   %-52s M%-4d K%-5d %5.0f/step %8.1f us %7.2f ms/step  %s %.3f' % (k['kernel'][:52], k['M'], k['K'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k['bound'], k['frac']))
