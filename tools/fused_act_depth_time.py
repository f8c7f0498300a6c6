"""Kernel time of the fused rollout step against network depth (1 / 2 / 3 / 5 hidden layers of 256, 107 observations, 90 actions, 4096 and
16 rows): run under `rocprofv3 --kernel-trace --output-format csv` and read the trace with tools/fused_act_depth_report.py (the launches
are too short for host-side timing: a Python call takes as long as the kernel).
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/fused_act_depth_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd.ppo import DiscreteFF

for n in (4096, 16):
    for hidden in [(256,), (256, 256), (256, 256, 256), (256,) * 5]:
        torch.manual_seed(1)
        pol = DiscreteFF(107, 90, hidden, "cuda:0")
        rows = pol.arena.stage_obs(torch.randn(n, 107, device="cuda").clamp_(-5, 5))
        q = torch.empty(n, 90, device="cuda").exponential_(1)
        for _ in range(100):
            pol.act_padded(rows, q)
        torch.cuda.synchronize()
