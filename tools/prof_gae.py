"""BASELINE configs[2] GAE scan (8192 x 256 steps) launched REPS times, ROTATING over bench.GAE_SETS independent input + output
sets (587 MB between two uses of a line: every scan is cold in the Infinity Cache and in L2, as in bench.py's timed region): the
target of `rocprofv3 --kernel-trace --stats` and of the FETCH_SIZE / WRITE_SIZE PMC passes (tools/round_profile.sh).
HOT=1: the rounds 1-3 form (one set re-scanned in place)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
fns, sets, n = bench.gae_sets(1 if os.environ.get("HOT") else bench.GAE_SETS)
reps = int(os.environ.get("REPS", 120))
for i in range(reps):
    fns[i % len(fns)]()
torch.cuda.synchronize()
