"""Device time of the rollout step's launches at 4096 x 107 (and a few other batch sizes): the fused kernel against the layer chain,
interleaved in one process (HIP events).  usage: python tools/act_kernel_time.py"""
import contextlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N

with contextlib.redirect_stdout(sys.stderr):
    learner, _ = bench.build_workload("cuda:0")
pol = learner.policy
L = N.lib()
for n in (64, 1024, 4096, 16384):
    obs = torch.randn(n, bench.OBS, device="cuda").clamp_(-5, 5)
    rows = pol.arena.stage_obs(obs)
    q = torch.empty(n, bench.ACT, device="cuda").exponential_(1)
    res = {}
    for rnd in range(3):
        for fused in (1, 0):
            N.check(L.rlppo_dbg_set(27, fused))
            res.setdefault(fused, []).append(bench.time_region(lambda: pol.act_padded(rows, q), 50, warm=5))
    N.check(L.rlppo_dbg_set(27, 1))
    print("n = %5d: fused launch %.4f ms, layer chain (4 GEMM launches + sampling) %.4f ms" % (n, np.median(res[1]), np.median(res[0])))
