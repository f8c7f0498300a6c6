"""What would ONE optimiser step of the 8-rank share (a 65,536-row pass + the fused clip + Adam) gain from being replayed as a hipGraph instead
of issued launch by launch?  The same library calls (rlppo_ppo_minibatch, rlppo_ppo_join, rlppo_clip_adam_pack2 with fixed arguments: the same
minibatch and step count every time -- a timing probe, not a training loop), issued eagerly and captured once + replayed.
usage: python tools/graph_step_probe.py"""
import contextlib, ctypes, os, sys, time

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    import bench
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.engine import stream_ptr, ptr
    import rlgym_ppo_amd.ppo.ppo_learner as PL
    L = N.lib()
    with contextlib.redirect_stdout(sys.stderr):
        learner, buf = bench.build_workload("cuda:0")
    PL.dist_info = lambda: (None, 0, 8)
    learner.learn(buf)                                     # warm: workspaces, packed copies, optimiser state
    torch.cuda.synchronize()
    args = learner._minibatch_args(buf, 0, 8)
    idx = buf.epoch_indices_device()
    args.slot, args.workspace = 0, learner._slot_ws[0]
    args.idx = idx.data_ptr()
    args.mb = learner._fused_rows
    args.mb_ratio = float(args.mb / learner.batch_size)
    dv = learner.value_optimizer.fused_descriptor(0.5)
    dp_ = learner.policy_optimizer.fused_descriptor(0.5)

    def step():
        st = stream_ptr()
        N.check(L.rlppo_ppo_minibatch(st, ctypes.byref(args)))
        N.check(L.rlppo_ppo_join(st))
        N.check(L.rlppo_clip_adam_pack2(st, ctypes.byref(dv), ctypes.byref(dp_), ptr(learner._opt_sync) if learner.one_launch_optimizer else None))

    def timed(fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    print("rows per pass %d" % args.mb)
    for rep in range(3):
        e = timed(step)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        r = timed(g.replay)
        print("one step issued launch by launch %.1f us; captured and replayed %.1f us (%.1f %%)" % (e, r, 100 * (r - e) / e))
