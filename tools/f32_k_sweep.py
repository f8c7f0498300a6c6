"""Time of the fp32 hidden-layer forward / dX product against K at M = 524,288, N = 256: intercept = what a launch spends outside
its K loop (prologue, epilogue, stores), slope = the MFMA-bound K loop.  usage: python tools/f32_k_sweep.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
M, Nn = 524288, 256
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
bias = torch.zeros(Nn, device="cuda")
C = torch.empty(M, Nn, device="cuda")
bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, Nn)), dtype=torch.uint8, device="cuda")
for epi, name in ((1, "forward + bitmask"), (3, "dX from bitmask")):
    row = []
    for K in (32, 64, 128, 256, 512):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(Nn, K, device="cuda") * 0.05
        fn = lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), K, P(W), K, P(bias) if epi == 1 else None, P(C), Nn, M, Nn, K, epi, P(bits)))
        row.append((K, bench.time_region(fn, 10, warm_s=0.2) * 1e3))
    print(name, "  ".join(f"K={k}: {us:6.1f} us" for k, us in row), "  (MFMA-only at 2.35 GHz: %.1f us per 32 of K)" % (2 * M * Nn * 32 / 154e12 * 1e6))
