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
        assert g.late and g.push, "this tool times the default form (host window + late noise)"
        steps = ("stage (control words + observations pushed)", "launch (graph replay)", "noise draw + push", "poll until done", "copy out")
        T = {k: [] for k in ("whole get_action", "python around ActGraph.run") + steps}
        m = n * g.q_per_row

        def stage():
            g.noise_seq = g.noise_seq % 0x7FFFFFFF + 1
            g._stage(g.ctl_arg, g.noise_seq, n, g.obs_arg, obs.ctypes.data, obs.nbytes)

        def publish():
            q = pol._draw_noise(n)
            g._push(g.q_arg, q.data_ptr(), 4 * m, g.ctl_arg + 8, g.noise_seq)

        for _ in range(400):
            t0 = pc(); pol.get_action(obs); t_all = pc() - t0
            T["whole get_action"].append(t_all)
            t0 = pc(); g.run(obs, None, n, pol._draw_bound(n), pol._verify); T["python around ActGraph.run"].append(t_all - (pc() - t0))
            t0 = pc(); stage(); t1 = pc(); value, count = g._launch(n); t2 = pc(); publish(); t3 = pc()
            rc = g._wait(g._done_ptr, count, value, 2000); t4 = pc()
            a, lp = torch.from_numpy(g.act_np[:n].copy()), torch.from_numpy(g.logp_np[:n].copy()); t5 = pc()
            assert rc == 0
            for k, v in zip(steps, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                T[k].append(v)
        print("n = %d:" % n, "; ".join("%s %.1f us" % (k, 1e6 * float(np.median(v))) for k, v in T.items()),
              "; polled %d, timeouts %d, second launches %d" % (g.polled, g.poll_timeouts, g.late_retries))
