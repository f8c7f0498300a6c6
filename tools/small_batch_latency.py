"""Wall-clock latency of DiscreteFF.get_action at the reference's process-per-environment batch sizes (8-80 observations per call,
batched_agent_manager.py:202-204) and a few larger ones: the hipGraph replay (ppo/_mlp.py::ActGraph), the eager one-launch step()
and -- RLPPO_TUNE-free, switched in process -- the layer chain.  Host observations and host noise (the bit-exact mode), results on
the host: the whole call as the reference's collector sees it.  usage: python tools/small_batch_latency.py"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N

with contextlib.redirect_stdout(sys.stderr):
    learner, _ = bench.build_workload("cuda:0")
pol = learner.policy
L = N.lib()


def wall(fn, reps=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


def parts(n, obs):
    """Where a graph-served call spends its time: the noise draw (behind the launch when the graph takes late noise), everything
    up to the wait, the whole call with the noise handed in."""
    g_ = pol._graphs[pol._graphs and sorted(k for k in pol._graphs if k >= n)[0]]
    t = {}
    t["noise draw"] = wall(lambda: pol._draw_noise(n))
    q = pol._draw_noise(n).clone()
    wait = g_._wait
    try:
        g_._wait = lambda *a: 0   # (the kernels of consecutive calls queue up on the stream; the last wall() call synchronises)
        t["stage + replay" + (" + publish" if g_.late else "") + " (no wait)"] = wall(lambda: g_.run(obs, q, n))
    finally:
        g_._wait = wait
    torch.cuda.synchronize()
    t["the call without its draw"] = wall(lambda: g_.run(obs, q, n))
    t["late noise"] = float(g_.late)
    return t


print("%6s %14s %14s %14s %14s" % ("n", "graph us", "step() us", "chain graph us", "chain eager us"))
for n in (1, 8, 32, 80, 256, 1024):
    obs = np.clip(np.random.RandomState(n).randn(n, bench.OBS), -5, 5).astype(np.float32)
    row = []
    for fused, graphs in ((1, True), (1, False), (0, True), (0, False)):
        N.check(L.rlppo_dbg_set(27, fused))
        pol.act_graphs = graphs
        if fused or graphs:
            row.append(wall(lambda: pol.get_action(obs)))
        else:  # the round-2 eager form: staged rows, layer chain, two read-backs
            def eager():
                a, lp = pol.act_padded(pol.arena.stage_obs(obs))
                return a.cpu(), lp.cpu()
            row.append(wall(eager))
    N.check(L.rlppo_dbg_set(27, 1))
    pol.act_graphs = True
    print("%6d %14.1f %14.1f %14.1f %14.1f" % (n, *row))
for n in (8, 80):
    obs = np.clip(np.random.RandomState(n).randn(n, bench.OBS), -5, 5).astype(np.float32)
    pol.get_action(obs)
    print("n = %d:" % n, ", ".join("%s %.1f us" % kv for kv in parts(n, obs).items()))
