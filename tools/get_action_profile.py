"""Where a small-batch DiscreteFF.get_action call (8-80 host observations, batched_agent_manager.py:202-204) spends its wall time:
the steps of ppo/_mlp.py::ActGraph.run timed one by one (median of 400 calls).  usage: python tools/get_action_profile.py"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N

with contextlib.redirect_stdout(sys.stderr):
    learner, _ = bench.build_workload("cuda:0")
pol = learner.policy
L = N.lib()
pc = time.perf_counter
for form in (0,):
    for n in (8, 80):
        obs = np.clip(np.random.RandomState(n).randn(n, bench.OBS), -5, 5).astype(np.float32)
        for _ in range(30):
            pol.get_action(obs)
        g = pol._graphs[(n + 15) // 16 * 16]
        T = {k: [] for k in ("whole get_action", "noise draw", "stage (2 numpy copies)", "launch (ctypes + hipLaunchKernel)", "poll until done",
                             "copy out", "same launch + stream.synchronize()")}
        for _ in range(400):
            t0 = pc(); pol.get_action(obs); T["whole get_action"].append(pc() - t0)
            t0 = pc(); q = pol._draw_noise(n); t1 = pc()
            g.obs_np[:n] = obs; g.q_np[:q.numel()] = q.reshape(-1).numpy(); t2 = pc()
            g.seq = g.seq % 0x7FFFFFFF + 1; g.opts.done_value = g.seq; g.body(); t3 = pc()
            rc = g._wait(g._done_ptr, (n + 15) // 16, g.seq, 2000); t4 = pc()
            a, lp = torch.from_numpy(g.act_np[:n].copy()), torch.from_numpy(g.logp_np[:n].copy()); t5 = pc()
            assert rc == 0
            for k, v in zip(list(T)[1:6], (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                T[k].append(v)
            t0 = pc(); g.seq = g.seq % 0x7FFFFFFF + 1; g.opts.done_value = g.seq; g.body(); torch.cuda.current_stream().synchronize()
            T["same launch + stream.synchronize()"].append(pc() - t0)
        print("n = %d:" % n,
              "; ".join("%s %.1f us" % (k, 1e6 * float(np.median(v))) for k, v in T.items()), "; polled %d, timeouts %d" % (g.polled, g.poll_timeouts))
