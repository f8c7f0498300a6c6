import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for dims in ([107,256,256,256,90], [107,256,256,256,1]):
    d = N.dims_array(dims); nl = len(dims)-1
    flat = torch.randn(int(L.rlppo_flat_floats(d, nl)), device="cuda") * 0.05
    packed = torch.zeros(int(L.rlppo_packed_floats(d, nl)), device="cuda")
    N.check(L.rlppo_net_pack(st(), d, nl, P(flat), P(packed)))
    obs = torch.randn(M, 128, device="cuda"); ldo = int(L.rlppo_padded_out(dims[-1]))
    out = torch.empty(M, ldo, device="cuda")
    ws = torch.empty(int(L.rlppo_forward_workspace_bytes(d, nl, M)), dtype=torch.uint8, device="cuda")
    fn = lambda: N.check(L.rlppo_mlp_forward(st(), d, nl, P(packed), P(obs), 128, M, 0, P(out), ldo, P(ws), ws.numel()))
    for fused in (1, 0, 1, 0):
        N.check(L.rlppo_dbg_set(6, fused))
        ms = np.median([bench.time_region(fn, 5) for _ in range(3)])
        fl = 2*M*sum(a*b for a,b in zip([128,256,256,256],[256,256,256,ldo]))
        print(dims[-1], "fused" if fused else "layer", f"{ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF")
