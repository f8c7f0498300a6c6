"""Launch time of the split-bf16 hidden forward / dX (csrc/gemm_split.hip) beside the fp32-MFMA kernels at the update's launch shape
(M = 524,288, 256 -> 256), and their error against float64.  usage: python tools/x3_kernel_time.py   (RLPPO_LIB picks a build)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
M, Nn, K = 524288, 256, 256
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, device="cuda", generator=g).clamp_(min=0) * torch.rand(M, K, device="cuda", generator=g)).contiguous()
W = ((torch.rand(Nn, K, device="cuda", generator=g) * 2 - 1) / 16).contiguous()
bias = (torch.rand(Nn, device="cuda", generator=g) - 0.5) * 0.1
planes = torch.zeros(3 * Nn * K, dtype=torch.bfloat16, device="cuda")
N.check(L.rlppo_dbg_pack_x3(st(), P(W), K, Nn, K, P(planes)))
C32, C3 = torch.empty(M, Nn, device="cuda"), torch.empty(M, Nn, device="cuda")
bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, Nn)), 8), dtype=torch.uint8, device="cuda")
rows = torch.arange(0, M, 131)[:4096].cuda()
truth = torch.relu(A[rows].double() @ W.double().t() + bias.double())
scale = truth.abs().max().item()
flop = 2 * M * Nn * K
for name, fn, out in (("fp32 MFMA fwd", lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), K, P(W), K, P(bias), P(C32), Nn, M, Nn, K, 1, P(bits))), C32),
                      ("split-bf16 fwd", lambda: N.check(L.rlppo_dbg_gemm_nt_x3(st(), P(A), K, P(planes), P(bias), P(C3), Nn, M, Nn, K, 0, P(bits))), C3),
                      ("fp32 MFMA dX ", lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), K, P(W), K, None, P(C32), Nn, M, Nn, K, 3, P(bits))), None),
                      ("split-bf16 dX ", lambda: N.check(L.rlppo_dbg_gemm_nt_x3(st(), P(A), K, P(planes), None, P(C3), Nn, M, Nn, K, 1, P(bits))), None)):
    fn()
    err = ""
    if out is not None:
        e = (out[rows].double() - truth).abs()
        err = f"err max {e.max().item() / scale:.2e} rms {e.pow(2).mean().sqrt().item() / scale:.2e} of max|C|"
    t = bench.time_region(fn, 20, warm_s=0.3) * 1e3
    print(f"{os.path.basename(os.environ.get('RLPPO_LIB', 'default'))} {name}  {t:7.1f} us  {flop / t / 1e6:6.1f} TFLOP/s (fp32-equivalent)  {err}", flush=True)
