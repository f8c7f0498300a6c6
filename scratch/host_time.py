"""How long does the HOST need to enqueue one learn() step, and how does that compare with the GPU time?  (8-GPU scaling:
each rank enqueues the same per-epoch host work but only 1/8 of the minibatches.)"""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
learner.n_epochs = 10
learner.learn(buf); torch.cuda.synchronize()
for world_emul in (1, 8):
    # emulate the per-rank minibatch share without process groups: monkeypatch slices_for_rank
    import rlgym_ppo_amd.ppo.ppo_learner as PL
    orig = PL.slices_for_rank
    PL.slices_for_rank = (lambda n, r, w: list(range(0, n, world_emul)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    learner.learn(buf)
    t_host = time.perf_counter() - t0          # includes the final stats .cpu() sync of learn()
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    PL.slices_for_rank = orig
    print(f"minibatch share 1/{world_emul}: learn() returned after {t_host*1e3:.1f} ms, GPU drained after {t_all*1e3:.1f} ms")
