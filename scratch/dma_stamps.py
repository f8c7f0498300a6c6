import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib(); raw = ctypes.CDLL(N.LIB_PATH); M = 65536
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A = torch.randn(M, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda"); C = torch.empty(M, 256, device="cuda")
buf = torch.zeros(1024 * 4 * 8, dtype=torch.int64, device="cuda")
fn = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, None, P(W), 256, P(b), None, 0, P(C), 256, M, 256, 256, 1))
bench.time_region(fn, 5, warm_s=0.3)
raw.rlppo_hack_set_stamp_buf.argtypes = [ctypes.c_void_p]
assert raw.rlppo_hack_set_stamp_buf(buf.data_ptr()) == 0
for _ in range(3): fn()
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(1024, 4, 8).astype(np.float64).mean(axis=(0, 1))
names = ["prologue", "DMA issue", "frags + MFMA", "wait DMA", "barrier", "epilogue", "TOTAL"]
print(" | ".join(f"{n} {v:.0f} ({100*v/a[6]:.1f}%)" for n, v in zip(names, a[:7])), "| 16 K-steps x 64 MFMA x 32 cycles =", 16*64*32)
