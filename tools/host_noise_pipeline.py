"""The bit-exact rollout noise pipeline (engine.HostExponential) at several look-ahead depths / helper-thread counts, on this host:
draws per ms in a tight loop, per 4096-agent get_action step, and per environment step of a whole collect (bench.iteration_leg).
usage: python tools/host_noise_pipeline.py [depth:threads ...]"""
import contextlib, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import numpy as np, torch
    import bench
    import rlgym_ppo_amd.engine as E
    torch.manual_seed(1)
    for _ in range(10):
        E.host_exponential((4096, 90))
    t0 = time.perf_counter()
    for _ in range(300):
        E.host_exponential((4096, 90))
    tight = (time.perf_counter() - t0) / 300 * 1e3
    E._HOST_EXP._drain()
    with contextlib.redirect_stdout(sys.stderr):
        learner, _ = bench.build_workload("cuda:0")
        obs = np.clip(np.random.RandomState(0).randn(4096, bench.OBS), -5, 5).astype(np.float32)
        for _ in range(20):
            learner.policy.get_action(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            learner.policy.get_action(obs)
        step = (time.perf_counter() - t0) / 200 * 1e3
        del learner
        r = bench.iteration_leg()
    print("depth %s threads %s: tight loop %.3f ms/draw, get_action step %.3f ms, collect %.1f ms (%.3f ms/env step), %.2f M agent-steps/s"
          % (os.environ["RLPPO_NOISE_LOOKAHEAD"], os.environ["RLPPO_NOISE_THREADS"], tight, step, r["collect_ms"], r["collect_ms_per_env_step"],
             r["steps_per_s"] / 1e6), flush=True)
    sys.exit(0)
for combo in (sys.argv[1:] or ["0:1", "1:1", "2:2", "3:2", "3:3", "4:2", "4:3"]):
    d, t = combo.split(":")
    env = dict(os.environ, RLPPO_NOISE_LOOKAHEAD=d, RLPPO_NOISE_THREADS=t)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout.strip() or out.stderr[-400:], flush=True)
