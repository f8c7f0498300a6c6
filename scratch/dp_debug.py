"""Repeat the 2-rank vs 1-rank update until the parameters disagree; print where and how."""
import os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.multiprocessing as mp
from test_gpu_dp import _build, _free_port


def worker(rank, world, port, out):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        learner, buf = _build()
    learner.learn(buf)
    opt = learner.policy_optimizer
    out[rank] = (learner.policy.arena.flat.cpu(), learner.value_net.arena.flat.cpu(), opt.exp_avg.cpu(), opt.exp_avg_sq.cpu())
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        learner, buf = _build()
    learner.learn(buf)
    ref_p, ref_v = learner.policy.arena.flat.cpu(), learner.value_net.arena.flat.cpu()
    opt = learner.policy_optimizer
    ref_m, ref_s = opt.exp_avg.cpu(), opt.exp_avg_sq.cpu()
    fails = 0
    for attempt in range(int(os.environ.get('ATTEMPTS', 12))):
        mgr = mp.Manager(); out = mgr.dict()
        mp.spawn(worker, args=(2, _free_port(), out), nprocs=2, join=True)
        p, v, m, s = out[0]
        dp, dv = (p - ref_p).abs(), (v - ref_v).abs()
        if dp.max() > 1e-5 or dv.max() > 1e-5:
            fails += 1
        if (dp.max() > 1e-5 or dv.max() > 1e-5) and os.environ.get("DETAIL"):
            bad = torch.nonzero(dp > 1e-5).flatten()
            print("policy params off:", bad.numel(), "of", p.numel(), "first idx", bad[:10].tolist())
            for i in bad[:8].tolist():
                print(f"  idx {i}: p {p[i]:.6e} ref {ref_p[i]:.6e} | m {m[i]:.3e} ref_m {ref_m[i]:.3e} | v {s[i]:.3e} ref_v {ref_s[i]:.3e}")
            badv = torch.nonzero(dv > 1e-5).flatten()
            print("critic params off:", badv.numel(), "of", v.numel(), badv[:10].tolist())
            break
    print('RLPPO_TUNE', os.environ.get('RLPPO_TUNE'), 'failures', fails, flush=True)
