"""DiscreteFF.get_action at 8 / 80 / 256 host observations in the forms of ppo/_mlp.py::ActGraph, one process per form (the environment
switches are read when a graph is built): default = host window + late noise; RLPPO_ACT_PUSH=0 = everything in pinned host memory (round 4's transport).  With late noise the kernel's own statistics are
printed: polls / wait of the first wave for its noise, and the timeline of the last workgroup (100 MHz wall clock).
usage: python tools/get_action_modes.py ; RLPPO_ACT_PUSH=0 python tools/get_action_modes.py"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd.ppo._mlp import _bucket
with contextlib.redirect_stdout(sys.stderr):
    learner, _ = bench.build_workload("cuda:0")
pol = learner.policy
def wall(fn, reps=600, warm=50):
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))
out = []
for n in (8, 80, 256):
    obs = np.clip(np.random.RandomState(n).randn(n, bench.OBS), -5, 5).astype(np.float32)
    out.append("%d: %.1f" % (n, wall(lambda: pol.get_action(obs))))
    g = pol._graphs[_bucket(n)]
    if g.late:
        st = []
        for _ in range(60):
            pol.get_action(obs)
            w = g.read_control_words()
            st.append([int(w[3]), int(w[4])] + [int(x) for x in w[8:12]])
        a_ = np.array(st[10:], dtype=np.int64)
        d = lambda i, j: np.median((a_[:, i] - a_[:, j]) & 0xFFFFFFFF) / 100
        out.append("[polls %.2f, wait %.2f us; last tile: start->obs staged %.1f, ->layers done %.1f, ->sampled %.1f]" % (a_[:, 0].mean(), a_[:, 1].mean() / 100, d(3, 2), d(4, 3), d(5, 4)))
print("push", os.environ.get("RLPPO_ACT_PUSH", "1"), " ".join(out))
