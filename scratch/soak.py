"""Two identical long runs of the cfg2 update (N learn() calls each) must end in bit-identical parameters and finite reports."""
import os, sys, contextlib, hashlib, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
digests = []
for run in range(2):
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        learner, buf = bench.build_workload("cuda:0")
        for _ in range(steps):
            rep = learner.learn(buf)
    torch.cuda.synchronize()
    p = torch.cat([learner.policy.arena.flat, learner.value_net.arena.flat]).cpu()
    assert torch.isfinite(p).all() and all(math.isfinite(float(v)) for v in rep.values())
    digests.append(hashlib.sha256(p.numpy().tobytes()).hexdigest())
    print(f"run {run}: {steps} learn() calls = {learner.cumulative_model_updates} optimiser steps, entropy {rep['Policy Entropy']:.4f}, "
          f"value loss {rep['Value Function Loss']:.4f}, sha256(params) {digests[-1][:16]}", flush=True)
print("bit-identical:", digests[0] == digests[1])
