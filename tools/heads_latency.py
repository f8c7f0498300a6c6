"""get_action at 8 / 80 host observations for the heads / widths the one-launch kernel does not cover (the layer chain as a hipGraph:
pad + forward layers + sampling + completion words).  RLPPO_TUNE="40=0": the forward layers as 128-row tiles (rounds 1-4) instead of one
wave per 16 x 16 output block; RLPPO_ACT_PUSH=0: inputs in pinned host memory.  usage: python tools/heads_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rlgym_ppo_amd.ppo import ContinuousPolicy, MultiDiscreteFF, DiscreteFF
def wall(fn, reps=400, warm=40):
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))
torch.manual_seed(0)
heads = {"gaussian 231 -> 512x4 -> 16": ContinuousPolicy(231, 16, (512, 512, 512, 512), "cuda:0"),
         "multi-discrete 107 -> 256x3": MultiDiscreteFF(107, (256, 256, 256), "cuda:0"),
         "discrete 107 -> 512x3 -> 90 (chain)": DiscreteFF(107, 90, (512, 512, 512), "cuda:0")}
for name, pol in heads.items():
    d = pol.arena.d_in
    out = []
    for n in (8, 80):
        obs = np.clip(np.random.RandomState(n).randn(n, d), -5, 5).astype(np.float32)
        out.append("n = %d: %.1f us" % (n, wall(lambda: pol.get_action(obs))))
    print("RLPPO_TUNE=%s RLPPO_ACT_PUSH=%s" % (os.environ.get("RLPPO_TUNE", ""), os.environ.get("RLPPO_ACT_PUSH", "1")), name, "; ".join(out))
