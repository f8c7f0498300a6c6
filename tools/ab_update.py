"""A/B of the round-3 changes to the PPO update, interleaved in ONE process on ONE device (cdna_hip_programming.md rule 24): the
1-rank learn() and the share of rank 0 of 8 (tools/rank_share.py's floor), with each change switched off in turn.
usage: python tools/ab_update.py [rounds]"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rlgym_ppo_amd.ppo.ppo_learner as PL
from rlgym_ppo_amd import _native as N

with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
L = N.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
configs = {
    "defaults": {},
    "paired launches at every size (dbg 29=2)": {"dbg": (29, 2)},
    "separate gather pass (dbg 26=0)": {"dbg": (26, 0)},
    "optimiser tail as fill+norms+update": {"one_launch": False},
    "one launch chain per network (dbg 29=0)": {"dbg": (29, 0)},
    "critic head backward not ordered after the policy loss (dbg 31=0)": {"dbg": (31, 0)},
    "critic head as its own matrix-vector launch (dbg 32=0)": {"dbg": (32, 0)},
    "paired launches stacked in z, not interleaved (dbg 33=0)": {"dbg": (33, 0)},
    "all off (round 2 launch structure)": {"dbg2": ((26, 0), (29, 0), (31, 0), (32, 0)), "one_launch": False},
}


def apply(cfg, on):
    for key, val in ([cfg["dbg"]] if "dbg" in cfg else []) + list(cfg.get("dbg2", ())):
        N.check(L.rlppo_dbg_set(key, val if on else 1))  # (1 = the default of every switch used here)
    learner.one_launch_optimizer = cfg.get("one_launch", True) if on else True


def timed(world, reps=4):
    PL.dist_info = lambda w=world: (None, 0, w)
    learner.learn(buf)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        learner.learn(buf)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


res = {(name, w): [] for name in configs for w in (1, 8)}
for _ in range(2):
    timed(1, 2)  # clock ramp
for r in range(rounds):
    for name, cfg in configs.items():
        apply(cfg, True)
        for w in (1, 8):
            res[(name, w)].append(timed(w))
        apply(cfg, False)
print("%-44s %22s %26s" % ("configuration (ms per 10-epoch learn(), median / min)", "1 rank", "share of rank 0 of 8"))
for name in configs:
    a, b = res[(name, 1)], res[(name, 8)]
    print("%-68s %8.2f / %8.2f      %8.2f / %8.2f" % (name, np.median(a), min(a), np.median(b), min(b)))
