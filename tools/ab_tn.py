"""A/B of gemm_tn: per-lane 64-bit pointers (gemm.hip) vs scalar-addressed buffer ops (gemm_sa.hip), interleaved."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536; dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A128, A256, A96, A32 = (torch.randn(M, k, device=dev) for k in (128, 256, 96, 32))
dW = torch.zeros(256 * 256, device=dev); db = torch.zeros(256, device=dev)
shapes = {"hidden 256x256": (A256, 256, A256, 256, 256, 256), "L0 256x107 (no gather)": (A256, 256, A128, 128, 256, 107),
          "head 90x256": (A96, 96, A256, 256, 90, 256), "vhead 1x256": (A32, 32, A256, 256, 1, 256)}
variants = {0: "vector-addr", 1: "scalar-addr", 2: "LDS-DMA TM32", 3: "LDS-DMA TM16"}
for rows in (512, 768, 1024):
  N.check(L.rlppo_dbg_set(2, rows)); print('rows per workgroup', rows)
  for name, (dY, ny, X, kx, out, in_) in shapes.items():
    fn = lambda: N.check(L.rlppo_dbg_gemm_tn(st(), P(dY), ny, ny, P(X), kx, None, kx, P(dW), P(db), out, in_, M))
    t = {v: [] for v in variants}
    for _ in range(5):
        for v in variants:
            N.check(L.rlppo_dbg_set(10, v)); t[v].append(bench.time_region(fn, 10))
    N.check(L.rlppo_dbg_set(10, 2))
    fl = 2 * M * ny * kx
    print("%-24s" % name + " | ".join("%s %6.1f us %6.1f TF" % (variants[v], np.median(t[v])*1e3, fl/np.median(t[v])/1e9) for v in variants))
