"""Reads a rocprofv3 --kernel-trace CSV and reports how the device time of a run divides into kernel-busy time (union over
streams) and idle gaps, and which kernels follow the longest gaps.  usage: python tools/trace_gaps.py <dir or kernel_trace.csv> [skip_fraction]"""
import csv, glob, os, sys
from collections import defaultdict

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5  # ignore the first part of the run (build, warm-up)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
rows = [r for r in rows if r[0] >= t0 + skip * (t1 - t0)]
span = rows[-1][1] - rows[0][0]
busy, end, gaps = 0, rows[0][0], defaultdict(lambda: [0, 0])
for s, e, name in rows:
    if s > end:
        g = gaps[name.split("(")[0][-60:]]
        g[0] += s - end
        g[1] += 1
        busy += e - s
    else:
        busy += max(0, e - end)
    end = max(end, e)
print(f"span {span / 1e6:.2f} ms, kernel-busy (union) {busy / 1e6:.2f} ms = {busy / span:.3f}, {len(rows)} launches")
for name, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  idle before {name:62s} {t / 1e3:9.1f} us in {n:5d} gaps = {t / max(n, 1) / 1e3:7.2f} us each")
