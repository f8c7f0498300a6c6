"""Time of the bf16 hidden-layer forward / dX product against K at M = 524,288, N = 512 (persistent 256 x 256, per-tile 256 x 256, 128 x 128): the intercept is what a
launch spends outside its K loop (prologue, epilogue, stores), the slope is the cost of a K step.  usage: python tools/b16_k_sweep.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
M, Nn = 524288, 512
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
bias = torch.zeros(Nn, device="cuda")
Cb = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, Nn)), dtype=torch.uint8, device="cuda")
for wide in (2, 1, 0):
    N.check(L.rlppo_dbg_set(23, wide))
    for mode, name in ((1, "forward"), (2, "dX")):
        row = []
        for K in (64, 128, 256, 512, 1024):
            A = torch.randn(M, K, device="cuda").bfloat16()
            W = (torch.randn(Nn, K, device="cuda") * 0.05).bfloat16()
            if mode == 1:
                fn = lambda: N.check(L.rlppo_dbg_gemm_nt_b16(st(), P(A), K, P(W), K, P(bias), None, 0, P(Cb), Nn, M, Nn, K, 1, 1, P(bits)))
            else:
                fn = lambda: N.check(L.rlppo_dbg_gemm_nt_b16(st(), P(A), K, P(W), K, None, None, 0, P(Cb), Nn, M, Nn, K, 3, 2, P(bits)))
            row.append((K, bench.time_region(fn, 10, warm_s=0.2) * 1e3))
        print({2: "256x256 persistent", 1: "256x256 per tile ", 0: "128x128          "}[wide], name, "  ".join(f"K={k}: {us:6.1f} us" for k, us in row))
N.check(L.rlppo_dbg_set(23, 2))
