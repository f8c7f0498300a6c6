"""Timeline summary of a rocprofv3 kernel trace: for the last `--epochs` optimiser steps (delimited by adam_kernel launches of
the policy net = every second adam launch), the span, the union of kernel intervals (GPU busy), the idle gaps and the time per
kernel.  usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [n_steps]"""
import csv, glob, sys
from collections import defaultdict
path = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
# an optimiser step ends with two adam launches (critic, policy); take the window between the last 2*n_steps+... launches
ends = adam[1::2]
lo, hi = ends[-n_steps - 1], ends[-1]
win = rows[lo + 1:hi + 1]
t0, t1 = win[0][0], max(r[1] for r in win)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = defaultdict(lambda: [0, 0])
for s, e, k in win:
    k = k.split("(")[0].replace("void ", "")
    per[k][0] += e - s
    per[k][1] += 1
print("steps %d  span %.3f ms/step  busy %.3f ms/step (%.1f%%)  sum of kernels %.3f ms/step" %
      (n_steps, (t1 - t0) / n_steps / 1e6, busy / n_steps / 1e6, 100 * busy / (t1 - t0), sum(v[0] for v in per.values()) / n_steps / 1e6))
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-58s %5.1f launches/step  %8.1f us/step  avg %7.1f us" % (k[:58], c / n_steps, t / n_steps / 1e3, t / c / 1e3))
