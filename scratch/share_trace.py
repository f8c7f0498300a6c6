"""One GPU doing the minibatch share of one rank of an 8-rank job (no collective), for rocprofv3 --kernel-trace:
python3 scratch/share_trace.py [world_emul]"""
import os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
world_emul = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
import rlgym_ppo_amd.ppo.ppo_learner as PL
PL.slices_for_rank = (lambda n, r, w: list(range(0, n // world_emul)))
learner.n_epochs = 10
for _ in range(4):
    learner.learn(buf)
torch.cuda.synchronize()
