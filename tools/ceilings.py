"""Measured-achievable ceilings of this box, beside the datasheet peaks the rooflines divide by: a large device copy
(HBM), large library GEMMs through PyTorch (rocBLAS / hipBLASLt: fp32 and bf16), and the register-only fp32 MFMA probe of
librlppo_diag.  usage: python tools/ceilings.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = "cuda"
src = torch.empty(1 << 30, dtype=torch.float32, device=dev)  # 4 GiB: far beyond the 256 MB memory-side cache
dst = torch.empty_like(src)
ms = bench.time_region(lambda: dst.copy_(src), 5, warm_s=0.3)
print(f"device copy 4 GiB -> 4 GiB: {ms:.3f} ms = {2 * src.numel() * 4 / ms / 1e9:.2f} TB/s of HBM traffic (read + write)")
ms = bench.time_region(lambda: dst.fill_(1.0), 5, warm_s=0.2)
print(f"fill 4 GiB (write only): {ms:.3f} ms = {src.numel() * 4 / ms / 1e9:.2f} TB/s")
ms = bench.time_region(lambda: src.sum(), 5, warm_s=0.2)
print(f"sum 4 GiB (read only): {ms:.3f} ms = {src.numel() * 4 / ms / 1e9:.2f} TB/s")
del src, dst
for dtype, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
    n = 8192
    a = torch.randn(n, n, device=dev, dtype=dtype)
    b = torch.randn(n, n, device=dev, dtype=dtype)
    torch.backends.cuda.matmul.allow_tf32 = False
    ms = bench.time_region(lambda: a @ b, 10, warm_s=0.5)
    print(f"library GEMM {name} {n}^3 (torch.matmul): {ms:.3f} ms = {2 * n ** 3 / ms / 1e9:.1f} TFLOP/s")
# the update's own shape through the library: M = 524,288, N = K = 256 (fp32)
a = torch.randn(524288, 256, device=dev)
w = torch.randn(256, 256, device=dev)
ms = bench.time_region(lambda: a @ w.T, 10, warm_s=0.5)
print(f"library GEMM fp32 524288 x 256 x 256 (no bias / ReLU / bitmask): {ms:.3f} ms = {2 * 524288 * 256 * 256 / ms / 1e9:.1f} TFLOP/s "
      f"(librlppo's forward with bias + ReLU + bitmask: see bench.py kernel_breakdown)")
# ... and the configs[4] hidden-layer shape, bf16 in / bf16 out, no bias / ReLU / bitmask (the bf16 update precision's forward)
ab = torch.randn(524288, 512, device=dev).bfloat16()
wb = (torch.randn(512, 512, device=dev) * 0.05).bfloat16()
ms = bench.time_region(lambda: ab @ wb.T, 10, warm_s=0.5)
print(f"library GEMM bf16 524288 x 512 x 512 (bf16 out, no epilogue): {ms:.3f} ms = {2 * 524288 * 512 * 512 / ms / 1e9:.1f} TFLOP/s "
      f"(librlppo's gemm_nt_b16w with bias + ReLU + rounding + bitmask: see bench.py --config cfg5 --precision bf16)")
del ab, wb
try:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import ctypes, _diag
    out = torch.empty(1024 * 256, device=dev)
    clocks = torch.zeros(2 * 1024, dtype=torch.int64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    iters = 4000
    fn = lambda: _diag.check(_diag.DL.rlppo_dbg_mfma_probe(st, ctypes.c_void_p(out.data_ptr()), 1024, iters, ctypes.c_void_p(clocks.data_ptr())))
    ms = bench.time_region(fn, 3, warm_s=0.5)
    flop = 1024 * 4 * iters * 64 * 2 * 16 * 16 * 4  # blocks x waves x iterations x 64 MFMAs per iteration x flop per 16x16x4 MFMA
    print(f"register-only fp32 MFMA probe: {flop / ms / 1e9:.1f} TFLOP/s")
except Exception as e:  # noqa: BLE001
    print("mfma probe skipped:", e)
