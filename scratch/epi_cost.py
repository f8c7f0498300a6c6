"""Cost of the epilogue flavours of gemm_nt at the hidden-layer shape: bias only (0), bias+ReLU (1), tanh (2), mask (3)."""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for M in (65536, 524288):
    A = torch.randn(M, 256, device="cuda"); A2 = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda")
    W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda"); C = torch.empty(M, 256, device="cuda")
    for K, Am in ((256, A), (128, A128)):
        res = {}
        for rep in range(2):
            for epi in (0, 1, 3):
                f = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(Am), K, None, P(W), K, P(b) if epi != 3 else None, P(A2) if epi == 3 else None, 256 if epi == 3 else 0, P(C), 256, M, 256, K, epi))
                res.setdefault(epi, []).append(bench.time_region(f, 20, warm_s=0.2) * 1e3)
        print(f"M={M} K={K}: " + "  ".join(f"epi {e}: {min(v):.1f} us" for e, v in res.items()), flush=True)
