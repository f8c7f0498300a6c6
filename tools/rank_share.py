"""Floor of an N-rank PPO update on ONE GPU: the learner evaluates only rank 0's share of every batch's minibatch slices (the
dealing of rlgym_ppo_amd/dp.py) and skips the collective -- what one rank of an N-rank job has to do per learn() apart from
the all-reduce.  Also times the two phases of the host permutation.  usage: python tools/rank_share.py [world ...]"""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rlgym_ppo_amd.ppo.ppo_learner as PL

with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
base = None
for world in ([int(w) for w in sys.argv[1:]] or (1, 2, 4, 8)):
    PL.dist_info = lambda w=world: (None, 0, w)
    for _ in range(2):
        learner.learn(buf)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        learner.learn(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    rate = 10 * bench.BATCH / dt
    base = base or rate
    print(f"share of rank 0 of {world}: {dt * 1e3:7.2f} ms per 10-epoch learn()  -> {world} such ranks: {rate / 1e6:6.1f} M samples/s "
          f"= {rate / base:.2f} x the 1-GPU rate (no collective)")
if os.environ.get("RANK_SHARE_PROFILE"):  # host-side profile of the last configuration (where does learn() spend host time?)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        learner.learn(buf)
    pr.disable()
    pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(22)
