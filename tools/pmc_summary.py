"""Condenses rocprofv3 --pmc counter_collection CSVs into one small per-kernel table (mean per dispatch).

Launches of one kernel with different grids get " [grid N]"; launches that differ only in their arguments can be told apart by
their position in a repeating launch cycle: PMC_CYCLE="<kernel name>=label0,label1,..." (tools/prof_kernels.py launches the dW
shapes in the order hidden, L0, head with the same grid)."""
import os
import csv
import glob
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(list))
rows_ = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        name = name[:name.index("(")] if "(" in name else name
        if not name.startswith(("void rlppo", "rlppo")):
            continue
        rows_.append((name.replace("void ", ""), row.get("Grid_Size", ""), row["Counter_Name"], float(row["Counter_Value"]),
                      int(row["Dispatch_Id"]), path))
cycles = {}
for item in filter(None, os.environ.get("PMC_CYCLE", "").split(";")):
    k, labels = item.split("=")
    cycles[k] = labels.split(",")
rows_.sort(key=lambda r: (r[5], r[4]))
seen = defaultdict(dict)  # (file, kernel) -> dispatch id -> ordinal
labelled = []
for name, grid, counter, value, disp, path in rows_:
    if name in cycles:
        order = seen[(path, name)]
        if disp not in order:
            order[disp] = len(order)
        name = name + " {" + cycles[name][order[disp] % len(cycles[name])] + "}"
    labelled.append((name, grid, counter, value))
rows_ = labelled
# one launch shape per line: a kernel launched with several grid sizes gets " [grid N]" appended (N = work-items)
grids = defaultdict(set)
for name, grid, _, _ in rows_:
    grids[name].add(grid)
for name, grid, counter, value in rows_:
    out[name + (" [grid %s]" % grid if len(grids[name]) > 1 else "")][counter].append(value)
counters = sorted({c for k in out.values() for c in k})
import csv as _csv, sys as _sys
w = _csv.writer(_sys.stdout)
w.writerow(["kernel"] + counters + ["dispatches"])
for k in sorted(out):
    n = max(len(v) for v in out[k].values())
    w.writerow([k] + ["%.6g" % (sum(out[k][c]) / len(out[k][c])) if out[k][c] else "" for c in counters] + [n])
