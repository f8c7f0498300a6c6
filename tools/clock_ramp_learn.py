"""Does learn() time depend on how long the GPU was busy before the timed region (clock ramp)?  The 10-epoch learn() of bench.py's workload at the
one-rank and the 8-rank-share size after 0 / 0.3 / 1 s of the same work and after 2 s of idling.  usage: python tools/clock_ramp_learn.py"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rlgym_ppo_amd.ppo.ppo_learner as PL
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
for world in (8, 1):
    PL.dist_info = lambda w=world: (None, 0, w)
    for warm_s in (0.0, 0.0, 0.3, 1.0, 0.0):
        t0 = time.perf_counter()
        learner.learn(buf); learner.learn(buf)
        while time.perf_counter() - t0 < warm_s:
            learner.learn(buf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            learner.learn(buf)
        torch.cuda.synchronize()
        print("world %d, busy %.1f s before the timed region: %.2f ms per learn()" % (world, warm_s, (time.perf_counter() - t) / 5 * 1e3), flush=True)
    time.sleep(2.0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        learner.learn(buf)
    torch.cuda.synchronize()
    print("world %d, after 2 s idle, no warm-up at all: %.2f ms" % (world, (time.perf_counter() - t) / 5 * 1e3), flush=True)
