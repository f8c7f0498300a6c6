import os, sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib(); M = 524288
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A256 = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda"); A96 = torch.randn(M, 96, device="cuda")
dW = torch.zeros(256 * 256, device="cuda"); db = torch.zeros(256, device="cuda")
for name, (dY, ny, X, kx, out, in_) in {"L0 256x107": (A256, 256, A128, 128, 256, 107), "L0 as 256x128": (A256, 256, A128, 128, 256, 128), "head 90x256": (A96, 96, A256, 256, 90, 256), "hidden": (A256, 256, A256, 256, 256, 256)}.items():
    res = []
    for rows in (512, 1024, 2048, 4096):
        N.check(L.rlppo_dbg_set(2, rows))
        ws = torch.empty(int(L.rlppo_dbg_gemm_tn_workspace_bytes(out, in_, M)), dtype=torch.uint8, device="cuda")
        fn = lambda: N.check(L.rlppo_dbg_gemm_tn_ws(st(), P(dY), ny, ny, P(X), kx, kx, P(dW), P(db), out, in_, M, P(ws), ws.numel()))
        res.append((rows, round(bench.time_region(fn, 10, warm_s=0.2) * 1e3)))
    print(name, res, flush=True)
