"""get_action latency by batch size (host obs in, host actions out; resident noise and host-drawn noise)."""
import os, sys, time, contextlib
sys.path.insert(0, ".")
import numpy as np, torch
from rlgym_ppo_amd.ppo import DiscreteFF
with contextlib.redirect_stdout(sys.stderr):
    pol = DiscreteFF(107, 90, (256, 256, 256), "cuda:0")
rs = np.random.RandomState(0)
for n in (8, 80, 512, 4096):
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    q = torch.empty(n, 90).exponential_(1).cuda()
    for name, fn in (("resident noise", lambda: pol.get_action(obs, noise=q)), ("host-drawn noise (bit-exact mode)", lambda: pol.get_action(obs))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        print(f"n={n:5d} {name:34s}: {dt*1e6:8.1f} us per call = {n/dt/1e6:7.3f} M obs/s", flush=True)
