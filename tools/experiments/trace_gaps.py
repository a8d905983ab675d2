"""Timeline of a rocprofv3 --kernel-trace run: per kernel name the launches, their mean duration and the mean gap in FRONT of them
(start - the previous kernel's end), over the last `frac` of the trace (steady state).

    python tools/experiments/trace_gaps.py <dir or kernel_trace.csv> [frac=0.5] [top=25]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:70]


def main():
    p = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    if os.path.isdir(p):
        p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = []
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[int(len(rows) * (1 - frac)):]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    busy = 0.0
    for i, (s, e, n) in enumerate(rows):
        dur[n] += e - s
        cnt[n] += 1
        busy += e - s
        if i:
            gap[n] += max(0, s - rows[i - 1][1])
    span = rows[-1][1] - rows[0][0]
    print("%s: %d launches, span %.3f ms, kernels busy %.3f ms (%.1f %%), gaps %.3f ms" %
          (os.path.basename(p), len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
    print("%-72s %7s %9s %9s %9s" % ("kernel", "n", "avg us", "gap us", "share %"))
    for n in sorted(dur, key=lambda k: -(dur[k] + gap[k]))[:top]:
        print("%-72s %7d %9.2f %9.2f %9.1f" % (n, cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3, 100.0 * (dur[n] + gap[n]) / span))


if __name__ == "__main__":
    main()
