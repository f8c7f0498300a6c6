"""A/B of a library switch on the reference's own configuration (buffer 150,000, B = MB = 50,000, 256x3, 10 epochs: bench.py's
ref_defaults leg), interleaved in one process.  usage: python tools/ab_ref_defaults.py KEY VALUE_A VALUE_B [KEY2 VALUE_A2 VALUE_B2 ...]"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N
from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner

L = N.lib()
n, B = bench.REF_BUFFER, bench.REF_BATCH
rs = np.random.RandomState(1)
obs = np.clip(rs.randn(n, bench.OBS), -5, 5).astype(np.float32)
z = np.zeros(n, np.float32)
torch.manual_seed(1)
with contextlib.redirect_stdout(sys.stderr):
    learner = PPOLearner(bench.OBS, bench.ACT, 0, bench.HID, bench.HID, (0.1, 1.0), B, 10, 3e-4, 3e-4, 0.2, 0.005, B, "cuda:0")
buf = ExperienceBuffer(n, 1, "cpu")
buf.submit_experience(obs, rs.randint(0, bench.ACT, n).astype(np.float32), (-np.log(bench.ACT) + 0.1 * rs.randn(n)).astype(np.float32), z, obs[:1].repeat(n, 0), z, z,
                      rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
trip = [tuple(int(x) for x in sys.argv[i:i + 3]) for i in range(1, len(sys.argv), 3)]
res = {"a": [], "b": []}
for rnd in range(4):
    for which in ("a", "b"):
        for key, va, vb in trip:
            N.check(L.rlppo_dbg_set(key, va if which == "a" else vb))
        learner.learn(buf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            learner.learn(buf)
        torch.cuda.synchronize()
        res[which].append((time.perf_counter() - t) / 3 * 1e3)
print("ref_defaults 10-epoch learn(): %s -> a: %.3f ms, b: %.3f ms  (runs %s / %s)" % (trip, np.median(res["a"]), np.median(res["b"]), ["%.2f" % x for x in res["a"]], ["%.2f" % x for x in res["b"]]))
