import os, sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
from rlgym_ppo_amd.util import torch_functions
L = N.lib()
rs = np.random.RandomState(0); n = 8192 * 256
d = lambda x: torch.as_tensor(x).cuda()
R, V = d(rs.randn(n).astype(np.float32)), d(rs.randn(n + 1).astype(np.float32))
D = d((rs.rand(n) < 0.005).astype(np.float32)); T = torch.zeros(n, device="cuda"); T[255::256] = 1
fn = lambda: torch_functions.gae_device(R, D, T, V, 0.99, 0.95, 1.7)
bench.time_region(fn, 1, warm_s=0.3)
cfgs = {"1 chunk/WG": (0, 1), "2 chunks/WG": (0, 2)}
t = {k: [] for k in cfgs}
for _ in range(7):
    for k, (dma, div) in cfgs.items():
        N.check(L.rlppo_dbg_set(18, div)); t[k].append(bench.time_region(fn, 20, warm=2))
for k in cfgs: print("%-20s %.2f us -> %.1f %% of 8 TB/s" % (k, np.median(t[k]) * 1e3, 28 * n / np.median(t[k]) / 1e6 / 8000 * 100))
