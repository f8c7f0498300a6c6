"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's algorithm for the hot path named in BASELINE.json
(rollout inference -> GAE -> PPO update of AechPro/rlgym-ppo v1.3.13).  It exists so that the
hand-written HIP path can be checked against something that (a) travels to the GPU box (the
reference's Python cannot) and (b) is itself pinned to the reference by the golden vectors in
tests/golden/ (generated here by importing the reference: tests/golden/make_golden.py).

Import rule (enforced by tests/test_layout.py): only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this package.  Nothing under rlgym_ppo_amd/ does; the
product path raises if its HIP extension is missing rather than falling back to this code.

Modules
    gae.py      ctypes front-end of gae_oracle.c (C restatement of compute_gae) + a pure-Python form
    nets.py     torch-CPU fp32 restatement of the three policy heads, the critic, and their sampling
    ppo.py      torch-CPU fp32 restatement of PPOLearner.learn + a float64 analytic-gradient form
    host.py     numpy restatement of ExperienceBuffer FIFO/shuffle and WelfordRunningStat
"""
