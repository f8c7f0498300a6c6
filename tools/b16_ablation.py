"""Launch time of the persistent bf16 hidden forward / dX (csrc/gemm_b16.hip) at configs[4]'s launch shape (M = 524,288, 512 -> 512) for ONE
library build (RLPPO_LIB): used with the -DB16_ABL=bits timing-only variants (parts of the kernel left out, results garbage).
usage: RLPPO_LIB=.../librlppo_bablN.so python tools/b16_ablation.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
M, Nn, K = 524288, 512, 512
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
bias = torch.zeros(Nn, device="cuda")
Cb = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, Nn)), dtype=torch.uint8, device="cuda")
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(Nn, K, device="cuda") * 0.05).bfloat16()
out = []
for mode, name in ((1, "forward"), (2, "dX")):
    if mode == 1:
        fn = lambda: N.check(L.rlppo_dbg_gemm_nt_b16(st(), P(A), K, P(W), K, P(bias), None, 0, P(Cb), Nn, M, Nn, K, 1, 1, P(bits)))
    else:
        fn = lambda: N.check(L.rlppo_dbg_gemm_nt_b16(st(), P(A), K, P(W), K, None, None, 0, P(Cb), Nn, M, Nn, K, 3, 2, P(bits)))
    fn()
    out.append(f"{name} {bench.time_region(fn, 20, warm_s=0.3) * 1e3:7.1f} us")
print(os.path.basename(os.environ.get("RLPPO_LIB", "default")), "  ".join(out), flush=True)
