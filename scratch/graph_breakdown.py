import os, sys, time, contextlib
sys.path.insert(0, ".")
import numpy as np, torch
from rlgym_ppo_amd.ppo import DiscreteFF
with contextlib.redirect_stdout(sys.stderr):
    pol = DiscreteFF(107, 90, (256, 256, 256), "cuda:0")
rs = np.random.RandomState(0)
for n in (80, 512, 1024):
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    q = torch.empty(n, 90).exponential_(1)
    pol.get_action(obs, noise=q)
    g = pol._graphs[max(pol._graphs)]
    T = [0, 0, 0, 0, 0]
    for _ in range(20):
        t0 = time.perf_counter(); qq = torch.empty(n, 90).exponential_(1)
        t1 = time.perf_counter(); g.obs_pin[:n].numpy()[...] = obs; g.q_pin[:n].copy_(qq)
        t2 = time.perf_counter(); g.graph.replay()
        t3 = time.perf_counter(); torch.cuda.current_stream().synchronize()
        t4 = time.perf_counter(); a = g.act_pin[:n].clone()
        t5 = time.perf_counter()
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[i] += d / 20
    print(f"n={n}: draw {T[0]*1e6:.0f} us, stage {T[1]*1e6:.0f} us, replay call {T[2]*1e6:.0f} us, sync {T[3]*1e6:.0f} us, clone {T[4]*1e6:.0f} us", flush=True)
