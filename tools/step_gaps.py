"""Steady-state timeline of the PPO update from a rocprofv3 --kernel-trace CSV: takes the optimiser steps between the last
`nsteps + 1` Adam launches, and reports per step the wall time, the time at least one kernel is running (union over streams), the
idle gaps (and which kernel follows them), and per-kernel busy time.  usage: python tools/step_gaps.py <dir or csv> [nsteps]"""
import csv, glob, os, sys
from collections import defaultdict

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
adam = [i for i, r in enumerate(rows) if "adam_fused_kernel" in r[2] or "adam_pack2_kernel" in r[2]]
lo, hi = adam[-nsteps - 1], adam[-1]
seg = rows[lo + 1:hi + 1]
t0, t1 = rows[lo][1], rows[hi][1]
busy, end = 0, t0
gaps = defaultdict(lambda: [0, 0])
per = defaultdict(lambda: [0, 0])
for s, e, name in seg:
    short = name.split("(")[0].replace("void ", "").replace("rlppo::", "")[:60]
    per[short][0] += e - s
    per[short][1] += 1
    if s > end:
        gaps[short][0] += s - end
        gaps[short][1] += 1
        busy += e - s
    else:
        busy += max(0, e - end)
    end = max(end, e)
span = t1 - t0
print(f"{nsteps} optimiser steps: {span / nsteps / 1e3:.1f} us per step, some kernel running {busy / span:.3f} of the time, "
      f"idle {(span - busy) / nsteps / 1e3:.1f} us per step, {len(seg) / nsteps:.1f} launches per step")
print("idle before (per step):")
for name, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {name:62s} {t / nsteps / 1e3:7.2f} us in {n / nsteps:5.2f} gaps")
print("kernel time (sum of durations, per step; kernels on different streams overlap):")
for name, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {name:62s} {t / nsteps / 1e3:8.2f} us in {n / nsteps:5.2f} launches = {t / n / 1e3:7.2f} us each")
