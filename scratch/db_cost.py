"""What does the bias-gradient column sum inside gemm_tn_dma cost?  Same launches with and without db."""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for M in (65536, 524288):
    A256 = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda"); A96 = torch.randn(M, 96, device="cuda")
    dW = torch.zeros(256 * 256, device="cuda"); db = torch.zeros(256, device="cuda")
    ws = torch.empty(max(int(L.rlppo_dbg_gemm_tn_workspace_bytes(o, i, M)) for o, i in ((256, 256), (256, 107), (90, 256))), dtype=torch.uint8, device="cuda")
    for name, (dY, ny, X, kx, out, in_) in {"hidden": (A256, 256, A256, 256, 256, 256), "L0": (A256, 256, A128, 128, 256, 107),
                                            "head": (A96, 96, A256, 256, 90, 256)}.items():
        res = []
        for use_db in (True, False, True, False):
            f = lambda: N.check(L.rlppo_dbg_gemm_tn_ws(st(), P(dY), ny, ny, P(X), kx, kx, P(dW), P(db) if use_db else None, out, in_, M, P(ws), ws.numel()))
            res.append(bench.time_region(f, 20, warm_s=0.2) * 1e3)
        print(f"M={M} dW {name}: with db {res[0]:.1f} / {res[2]:.1f} us, without {res[1]:.1f} / {res[3]:.1f} us", flush=True)
