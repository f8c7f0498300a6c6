"""Does the ~100 TFLOP/s plateau of every GEMM variant come from the clock the chip holds on real data?  Same staged
kernel, same shape, operands: N(0,1) random vs constant vs zero (MI355X_MICROARCH.md 'DVFS give-back')."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536; dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
bias = torch.zeros(256, device=dev); C = torch.empty(M, 256, device=dev)
cases = {"randn": (torch.randn(M, 256, device=dev), torch.randn(256, 256, device=dev) * 0.05),
         "const": (torch.full((M, 256), 0.5, device=dev), torch.full((256, 256), 0.25, device=dev)),
         "zeros": (torch.zeros(M, 256, device=dev), torch.zeros(256, 256, device=dev)),
         "small ints": (torch.randint(0, 4, (M, 256), device=dev).float(), torch.randint(0, 4, (256, 256), device=dev).float())}
for rnd in range(2):
    for name, (A, W) in cases.items():
        fn = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, None, P(W), 256, P(bias), None, 0, P(C), 256, M, 256, 256, 1))
        ms = np.median([bench.time_region(fn, 20) for _ in range(3)])
        print(f"{name:10s}: {ms*1e3:7.1f} us  {2*M*256*256/ms/1e9:6.1f} TF")
