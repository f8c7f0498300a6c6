import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536; dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A256 = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.05; bias = torch.randn(256, device=dev)
C = torch.empty(M, 256, device=dev)
fn = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A256), 256, None, P(W), 256, P(bias), None, 0, P(C), 256, M, 256, 256, 1))
N.check(L.rlppo_dbg_set(6, 0))
for sg in (0, 1, 2, 4, 6, 8, 12, 16, 0):
    N.check(L.rlppo_dbg_set(7, sg))
    ms = np.median([bench.time_region(fn, 10) for _ in range(3)])
    print(f"stagger {sg:2d} x512 cycles: {ms*1e3:7.1f} us  {2*M*256*256/ms/1e9:6.1f} TF")
