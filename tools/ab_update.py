"""A/B of whole PPOLearner.learn() (cfg2 workload) under the library's tuning switches, interleaved in one process."""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
learner.n_epochs = 3
configs = {"dw streams": (1, ((4, 1), (6, 0), (8, 1))), "no dw streams": (1, ((4, 1), (6, 0), (8, 0))), "dw streams, 1 chain stream": (1, ((4, 0), (6, 0), (8, 1)))}
res = {k: [] for k in configs}
learner.learn(buf)
for rnd in range(3):
    for name, (slots, sets) in configs.items():
        learner.n_slots = slots
        for k, v in sets:
            N.check(L.rlppo_dbg_set(k, v))
        torch.cuda.synchronize(); t = time.perf_counter(); learner.learn(buf); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t) / learner.n_epochs)
for name, v in res.items():
    print(f"{name}: {np.median(v)*1e3:7.2f} ms/epoch  {524288/np.median(v)/1e6:6.2f} M samples/s")
