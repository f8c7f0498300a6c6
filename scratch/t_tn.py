import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, ".")
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A256 = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda"); A96 = torch.randn(M, 96, device="cuda")
dW = torch.zeros(256*256, device="cuda"); db = torch.zeros(256, device="cuda")
shapes = {"hidden 256x256": (A256,256,A256,256,256,256), "L0 256x107": (A256,256,A128,128,256,107), "head 90x256": (A96,96,A256,256,90,256)}
for name,(dY,ny,X,kx,out,in_) in shapes.items():
    fn = lambda: N.check(L.rlppo_dbg_gemm_tn(st(), P(dY), ny, ny, P(X), kx, None, kx, P(dW), P(db), out, in_, M))
    res = []
    for rows in (128, 256, 384, 512, 768, 1024):
        N.check(L.rlppo_dbg_set(2, rows))
        res.append((rows, round(bench.time_region(fn, 20, warm_s=0.2)*1e3, 1)))
    print(sys.argv[1], name, res, flush=True)
N.check(L.rlppo_dbg_set(2, 0))
for name,(dY,ny,X,kx,out,in_) in shapes.items():
    ws = torch.empty(int(L.rlppo_dbg_gemm_tn_workspace_bytes(out, in_, M)), dtype=torch.uint8, device="cuda")
    fn = lambda: N.check(L.rlppo_dbg_gemm_tn_ws(st(), P(dY), ny, ny, P(X), kx, kx, P(dW), P(db), out, in_, M, P(ws), ws.numel()))
    print("partial tiles + reduce", name, round(bench.time_region(fn, 20, warm_s=0.2)*1e3, 1), "us; ws MB", ws.numel()/1e6, flush=True)
