"""Where a rollout step (4096 x 107 observations -> actions / log-probs on the host) spends its time: the host -> device hand-over of
the observations (pageable numpy, pinned numpy, already resident), the launch chain on the device (HIP events), the read-back.
usage: python tools/rollout_breakdown.py"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

with contextlib.redirect_stdout(sys.stderr):
    learner, _ = bench.build_workload("cuda:0")
pol = learner.policy
n, d, A = bench.N_AGENTS, bench.OBS, bench.ACT
rs = np.random.RandomState(0)
obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
obs_pin_t = torch.from_numpy(obs).pin_memory()
obs_pin = obs_pin_t.numpy()
obs_dev = torch.from_numpy(obs).cuda()
q_host = torch.empty(n, A).exponential_(1)
q_pin = q_host.pin_memory()
q_dev = q_host.cuda()


def wall(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def dev(fn, reps=50):
    return bench.time_region(fn, reps, warm=5)


rows = pol.arena.stage_obs(obs_dev)
print("H2D obs pageable numpy -> device      %.4f ms" % wall(lambda: torch.from_numpy(obs).to("cuda")))
print("H2D obs pinned numpy -> device        %.4f ms" % wall(lambda: torch.from_numpy(obs_pin).to("cuda", non_blocking=True)))
print("H2D noise pageable -> device          %.4f ms" % wall(lambda: q_host.to("cuda")))
print("H2D noise pinned -> device            %.4f ms" % wall(lambda: q_pin.to("cuda", non_blocking=True)))
print("stage_obs (pad kernel), device input  %.4f ms wall, %.4f ms device" % (wall(lambda: pol.arena.stage_obs(obs_dev)), dev(lambda: pol.arena.stage_obs(obs_dev))))
print("act_padded (forward + sample), device %.4f ms wall, %.4f ms device" % (wall(lambda: pol.act_padded(rows, q_dev)), dev(lambda: pol.act_padded(rows, q_dev))))
a, lp = pol.act_padded(rows, q_dev)
print("D2H actions + logp (.cpu() x2)        %.4f ms" % wall(lambda: (a.cpu(), lp.cpu())))
print("get_action(obs pageable, noise dev)   %.4f ms" % wall(lambda: pol.get_action(obs, noise=q_dev)))
print("get_action(obs pinned,   noise dev)   %.4f ms" % wall(lambda: pol.get_action(obs_pin, noise=q_dev)))
print("get_action(obs device,   noise dev)   %.4f ms" % wall(lambda: pol.get_action(obs_dev, noise=q_dev)))
print("get_action(obs pageable, host noise)  %.4f ms" % wall(lambda: pol.get_action(obs)))
print("get_action(obs pinned,   host noise)  %.4f ms" % wall(lambda: pol.get_action(obs_pin)))
from rlgym_ppo_amd.engine import host_exponential
print("host_exponential((4096, 90)) alone    %.4f ms" % wall(lambda: host_exponential((n, A))))
