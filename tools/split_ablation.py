"""Launch time of the persistent split-bf16 kernel (csrc/gemm_split.hip) at the update's launch shape (M = 524,288, 256 -> 256) for
ONE library build; tools/split_ablation.sh builds the -DSPLIT_ABL=bits variants (parts of a K step left out, results garbage) and
runs this under each.  usage: [MODE=0|1] RLPPO_LIB=.../librlppo_ablN.so python tools/split_ablation.py   (MODE 1 = dX, 0 = forward)"""
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
C3 = torch.empty(M, Nn, device="cuda")
bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, Nn)), 8), dtype=torch.uint8, device="cuda")
mode = int(os.environ.get("MODE", "1"))
fn = lambda: N.check(L.rlppo_dbg_gemm_nt_x3(st(), P(A), K, P(planes), None if mode else P(bias), P(C3), Nn, M, Nn, K, mode, P(bits)))
fn()
t = bench.time_region(fn, 20, warm_s=0.3) * 1e3
print(os.path.basename(os.environ.get("RLPPO_LIB", "default")), f"mode {mode} {t:7.1f} us", flush=True)
