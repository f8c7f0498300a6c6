"""Kernel time of the fused rollout step against network depth (1 / 2 / 3 / 5 hidden layers of 256, 107 observations, 90 actions, 4096 and
16 rows): run under `rocprofv3 --kernel-trace --output-format csv` and read the trace (the launches are too short for host-side
timing: a Python call takes as long as the kernel).  usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/fused_act_depth_time.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_kernels as T
from oracle import nets
from rlgym_ppo_amd import _native as N
L = N.lib()
for n in (4096, 16):
    for hidden in [(256,), (256, 256), (256, 256, 256), (256,) * 5]:
        torch.manual_seed(1)
        d, A = 107, 90
        net = T.Net(L, nets.init_mlp(d, hidden, A))
        rs = np.random.RandomState(d)
        rows = net.pad(np.clip(rs.randn(n, d) * 2, -5, 5).astype(np.float32))
        q = T.dev(rs.exponential(size=(n, A)).astype(np.float32))
        act = torch.empty(n, dtype=torch.int64, device="cuda"); logp = torch.empty(n, device="cuda"); w = net.ws(n)
        for _ in range(100):
            T.check(L, L.rlppo_discrete_act(T.stream(), net.dims_c, net.nl, T.P(net.packed), T.P(rows), net.ld_in, n, T.P(q), T.P(act), T.P(logp), None, T.P(w), w.numel()))
        torch.cuda.synchronize()
