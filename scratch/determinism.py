"""Run the DP test's single-process update several times from identical initial state; report run-to-run differences."""
import os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_dp import _build

res = []
for r in range(6):
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        learner, buf = _build()
    learner.learn(buf)
    res.append((learner.policy.arena.flat.cpu().clone(), learner.value_net.arena.flat.cpu().clone()))
for r in range(1, 6):
    dp = (res[r][0] - res[0][0]).abs(); dv = (res[r][1] - res[0][1]).abs()
    print(f"run {r} vs 0: policy max diff {dp.max():.3e} (rel {dp.max()/res[0][0].abs().max():.2e}, {int((dp>1e-5).sum())} params >1e-5), "
          f"critic {dv.max():.3e} (rel {dv.max()/res[0][1].abs().max():.2e}, {int((dv>1e-5).sum())} params >1e-5)")
