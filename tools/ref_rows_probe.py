"""learn() at B = MB = rows for a list of row counts (buffer 3 x rows, 256x3, the reference's defaults otherwise): ms per optimiser step and
samples/s at 10 epochs and at 1 epoch -- how the step time depends on the number of row tiles of a pass (a 50,000-row pass is 782 workgroups
per hidden launch on 1024 slots).  usage: python tools/ref_rows_probe.py [rows ...]"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner

for B in ([int(x) for x in sys.argv[1:]] or [49152, 50000, 65536]):
    n = 3 * B
    rs = np.random.RandomState(1)
    obs = np.clip(rs.randn(n, bench.OBS), -5, 5).astype(np.float32)
    z = np.zeros(n, np.float32)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(sys.stderr):
        learner = PPOLearner(bench.OBS, bench.ACT, 0, bench.HID, bench.HID, (0.1, 1.0), B, 10, 3e-4, 3e-4, 0.2, 0.005, B, "cuda:0")
    buf = ExperienceBuffer(n, 1, "cpu")
    buf.submit_experience(obs, rs.randint(0, bench.ACT, n).astype(np.float32), (-np.log(bench.ACT) + 0.1 * rs.randn(n)).astype(np.float32), z, obs[:1].repeat(n, 0), z, z,
                          rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
    line = "rows %6d:" % B
    for epochs, reps in ((10, 5), (1, 20)):
        learner.n_epochs = epochs
        for _ in range(2):
            learner.learn(buf)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            learner.learn(buf)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        dt = float(np.median(ts))
        line += "  %2d epoch(s): %.3f ms per learn(), %.4f ms per step, %.2f M samples/s" % (epochs, dt * 1e3, dt * 1e3 / (3 * epochs), 3 * B * epochs / dt / 1e6)
    print(line, flush=True)
    del learner, buf
    torch.cuda.empty_cache()
