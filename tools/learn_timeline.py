"""Host timeline of ONE learn() at the reference's own configuration (buffer 150,000, batch = minibatch = 50,000, 1 epoch = 3 optimiser
steps): when, after learn() was entered, each library call was made and returned, and when the report's completion word arrived --
what the ~0.3 ms of a 3 ms learn() that is not its three steps is spent on.  usage: python tools/learn_timeline.py [epochs]"""
import contextlib, os, sys, time

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    import bench
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B, n = bench.REF_BATCH, bench.REF_BUFFER
    torch.manual_seed(1)
    with contextlib.redirect_stdout(sys.stderr):
        learner = PPOLearner(bench.OBS, bench.ACT, 0, bench.HID, bench.HID, (0.1, 1.0), B, epochs, 3e-4, 3e-4, 0.2, 0.005, B, "cuda:0")
    rs = np.random.RandomState(0)
    obs = np.clip(rs.randn(n, bench.OBS), -5, 5).astype(np.float32)
    buf = ExperienceBuffer(n, 1, "cpu")
    z = np.zeros(n, np.float32)
    buf.submit_experience(obs, rs.randint(0, bench.ACT, n).astype(np.float32), -4.5 + 0.1 * rs.randn(n).astype(np.float32), z, obs, z, z,
                          rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
    L = N.lib()
    log = []

    def wrap(obj, name, label=None):
        fn = getattr(obj, name)

        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            log.append((label or name, t, time.perf_counter()))
            return r
        setattr(obj, name, w)
    for name in ("rlppo_ppo_minibatch", "rlppo_ppo_join", "rlppo_clip_adam_pack2", "rlppo_learn_report", "rlppo_host_wait_words"):
        wrap(L, name)
    wrap(buf, "epoch_indices_device")
    wrap(learner, "_minibatch_args")
    wrap(buf._perm, "take", "  perm.take")
    wrap(buf._ring, "take", "  ring.take")
    wrap(buf, "refill_shuffle")
    wrap(learner.policy.arena, "ensure_packed", "pol.ensure_packed")
    wrap(learner.value_net.arena, "ensure_packed", "val.ensure_packed")
    for _ in range(10):
        learner.learn(buf)
    torch.cuda.synchronize()
    rows = []
    for _ in range(30):
        del log[:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        learner.learn(buf)
        t1 = time.perf_counter()
        rows.append(([(nm, (a - t0) * 1e6, (b - t0) * 1e6) for nm, a, b in log], (t1 - t0) * 1e6))
    rows.sort(key=lambda r: r[1])
    ev, total = rows[len(rows) // 2]
    print("median learn(): %.1f us (%d epoch(s), %d optimiser steps); its calls (start -> end, us after entry):" % (total, epochs, 3 * epochs))
    for nm, a, b in ev[:16] + ([("...", 0, 0)] if len(ev) > 32 else []) + (ev[-16:] if len(ev) > 32 else ev[16:]):
        print("  %-28s %9.1f -> %9.1f  (%.1f)" % (nm, a, b, b - a))
    print("  %-28s %9.1f" % ("learn() returns", total))
