"""EXPERIMENT: the 256 -> 256 hidden forward with fp32 data and split-bf16 products (csrc/diag/split_probe.hip)
beside the product's fp32-MFMA kernel (gemm_nt_dma_kernel through rlppo_dbg_gemm_nt_bits): launch time at M = 524,288 and error against
float64 truth on a row sample, for 6 / 4 / 3 / 1 piece products.  usage: python tools/split_bf16_probe.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench, _diag
from rlgym_ppo_amd import _native as N
L = N.lib()
D = _diag.DL
M, Nn, K = 524288, 256, 256
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, device="cuda", generator=g).clamp_(min=0) * torch.rand(M, K, device="cuda", generator=g)).contiguous()   # relu-like activations
W = ((torch.rand(Nn, K, device="cuda", generator=g) * 2 - 1) / 16).contiguous()
bias = (torch.rand(Nn, device="cuda", generator=g) - 0.5) * 0.1


def split_planes(w, rne):
    """fp32 [N][K] -> uint16 [K / 32][3][N][32]: three bf16 pieces, by truncation (w = h + m + l exactly) or rounded to nearest even
    (|m| <= 2^-9 |w|, |l| <= 2^-18 |w|), stage-major."""
    if rne:
        h = w.bfloat16().float()
        r = w - h
        m = r.bfloat16().float()
        r2 = r - m
        l = r2.bfloat16().float()
    else:
        h = (w.contiguous().view(torch.int32) & -65536).view(torch.float32)
        r = w - h
        m = (r.view(torch.int32) & -65536).view(torch.float32)
        r2 = r - m
        l = (r2.view(torch.int32) & -65536).view(torch.float32)
        assert torch.equal(h + m + l, w)
    planes = torch.stack([(x.contiguous().view(torch.int32) >> 16).to(torch.int16) for x in (h, m, l)])        # [3][N][K]
    return planes.view(3, Nn, K // 32, 32).permute(2, 0, 1, 3).contiguous()


Wsplit = {False: split_planes(W, False), True: split_planes(W, True)}
C32 = torch.empty(M, Nn, device="cuda")
Cs = torch.empty(M, Nn, device="cuda")
bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, Nn)), 8), dtype=torch.uint8, device="cuda")
f32 = lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), K, P(W), K, P(bias), P(C32), Nn, M, Nn, K, 1, P(bits)))
rows = torch.arange(0, M, 131)[:4096].cuda()
truth = torch.relu(A[rows].double() @ W.double().t() + bias.double())
scale = truth.abs().max().item()
f32()
e32 = (C32[rows].double() - truth).abs()
t32 = bench.time_region(f32, 20, warm_s=0.3) * 1e3
flop = 2 * M * Nn * K
print(f"fp32 MFMA kernel (product, + bitmask)   {t32:7.1f} us  {flop / t32 / 1e6:6.1f} TFLOP/s   err max {e32.max().item() / scale:.2e}  rms {e32.pow(2).mean().sqrt().item() / scale:.2e} of max|C|")
for rne in (True, False):
    Ws = Wsplit[rne]
    for terms in ((206, 8, 6, 4, 3) if rne else (8, 6, 4, 3, 1)):
        for store in (1, 0):
            code = terms if terms > 200 else terms + (100 if rne else 0)
            fn = lambda: _diag.check(D.rlppo_dbg_gemm_nt_split(st(), P(A), K, P(Ws), P(bias), P(Cs), Nn, M, Nn, K, code, store))
            if store:
                Cs.zero_()
                fn()
                es = (Cs[rows].double() - truth).abs()
                err = f"err max {es.max().item() / scale:.2e} ({es.max().item() / e32.max().item():.2f} x fp32 MFMA)  rms {es.pow(2).mean().sqrt().item() / scale:.2e} ({(es.pow(2).mean().sqrt() / e32.pow(2).mean().sqrt()).item():.2f} x)"
            t = bench.time_region(fn, 20, warm_s=0.3) * 1e3
            print(f"split-bf16 {'nearest   ' if rne else 'truncation'} {'6 (small ones summed apart)' if terms > 200 else terms} piece products, {'stores     ' if store else 'K loop only'} {t:7.1f} us  {flop / t / 1e6:6.1f} TFLOP/s (fp32-equivalent)   {err if store else ''}", flush=True)
