"""A/B of a library switch (rlppo_dbg_set key) on the N-rank share of the update (tools/rank_share.py's measurement), interleaved in
one process.  usage: python tools/ab_rank_share.py KEY VALUE_A VALUE_B [world ...]"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rlgym_ppo_amd.ppo.ppo_learner as PL
from rlgym_ppo_amd import _native as N

key, va, vb = (int(x) for x in sys.argv[1:4])
worlds = [int(w) for w in sys.argv[4:]] or [8, 4, 1]
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
L = N.lib()
for world in worlds:
    PL.dist_info = lambda w=world: (None, 0, w)
    res = {va: [], vb: []}
    for rnd in range(4):
        for v in (va, vb):
            N.check(L.rlppo_dbg_set(key, v))
            learner.learn(buf)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                learner.learn(buf)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / 5 * 1e3)
    print(f"share of rank 0 of {world}: key {key} = {va}: {np.median(res[va]):7.3f} ms   = {vb}: {np.median(res[vb]):7.3f} ms per 10-epoch learn()"
          f"   (runs {['%.2f' % x for x in res[va]]} / {['%.2f' % x for x in res[vb]]})")
