"""Condense the kernel trace of tools/fused_act_depth_time.py.  usage: python tools/fused_act_depth_report.py <kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "discrete_act_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i = 0
for n in (4096, 16):
    for nh in (1, 2, 3, 5):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows[i:i + 100]]
        i += 100
        print("%5d rows, %d hidden layers of 256: median %.2f us (min %.2f)" % (n, nh, np.median(d), np.min(d)))
