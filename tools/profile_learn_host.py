"""cProfile of the HOST side of one learn() at the reference's own configuration (buffer 150,000, batch = minibatch = 50,000, 1 epoch:
3 optimiser steps, ~2.7 ms of GPU work): what Python and the launches cost around it.  usage: python tools/profile_learn_host.py"""
import cProfile, contextlib, io, os, pstats, sys, time

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    import bench
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    B, n = bench.REF_BATCH, bench.REF_BUFFER
    torch.manual_seed(1)
    with contextlib.redirect_stdout(sys.stderr):
        learner = PPOLearner(bench.OBS, bench.ACT, 0, bench.HID, bench.HID, (0.1, 1.0), B, 1, 3e-4, 3e-4, 0.2, 0.005, B, "cuda:0")
    rs = np.random.RandomState(0)
    obs = np.clip(rs.randn(n, bench.OBS), -5, 5).astype(np.float32)
    buf = ExperienceBuffer(n, 1, "cpu")
    z = np.zeros(n, np.float32)
    buf.submit_experience(obs, rs.randint(0, bench.ACT, n).astype(np.float32), -4.5 + 0.1 * rs.randn(n).astype(np.float32), z, obs, z, z,
                          rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
    for _ in range(5):
        learner.learn(buf)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); learner.learn(buf); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("learn(): median %.3f ms, min %.3f ms" % (1e3 * np.median(ts), 1e3 * min(ts)))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        learner.learn(buf)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print("\n".join(l[:150] for l in s.getvalue().splitlines() if l.strip()))
