import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for M in (65536, 131072, 524288):
    A256 = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda"); A96 = torch.randn(M, 96, device="cuda")
    W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda"); C = torch.empty(M, 256, device="cuda"); C96 = torch.empty(M, 96, device="cuda")
    dW = torch.zeros(256 * 256, device="cuda"); db = torch.zeros(256, device="cuda")
    ws = torch.empty(max(int(L.rlppo_dbg_gemm_tn_workspace_bytes(o, i, M)) for o, i in ((256, 256), (256, 107), (90, 256))), dtype=torch.uint8, device="cuda")
    kern = {
        "fwd hidden": (lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A256), 256, None, P(W), 256, P(b), None, 0, P(C), 256, M, 256, 256, 1)), 2 * M * 256 * 256),
        "fwd L0": (lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A128), 128, None, P(W), 128, P(b), None, 0, P(C), 256, M, 256, 128, 1)), 2 * M * 256 * 128),
        "fwd head 96": (lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A256), 256, None, P(W), 256, P(b), None, 0, P(C96), 96, M, 96, 256, 0)), 2 * M * 96 * 256),
        "dX hidden": (lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A256), 256, None, P(W), 256, None, P(A256), 256, P(C), 256, M, 256, 256, 3)), 2 * M * 256 * 256),
        "dX head 96->256": (lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A96), 96, None, P(W), 96, None, P(A256), 256, P(C), 256, M, 256, 96, 3)), 2 * M * 256 * 96),
        "dW hidden": (lambda: N.check(L.rlppo_dbg_gemm_tn_ws(st(), P(A256), 256, 256, P(A256), 256, 256, P(dW), P(db), 256, 256, M, P(ws), ws.numel())), 2 * M * 256 * 256),
        "dW L0": (lambda: N.check(L.rlppo_dbg_gemm_tn_ws(st(), P(A256), 256, 256, P(A128), 128, 128, P(dW), P(db), 256, 107, M, P(ws), ws.numel())), 2 * M * 256 * 128),
        "dW head": (lambda: N.check(L.rlppo_dbg_gemm_tn_ws(st(), P(A96), 96, 96, P(A256), 256, 256, P(dW), P(db), 90, 256, M, P(ws), ws.numel())), 2 * M * 128 * 256),
    }
    print("M =", M, " | ".join("%s %.0f us %.0f TF" % (k, bench.time_region(f, 10, warm_s=0.2) * 1e3, fl / bench.time_region(f, 10) / 1e9) for k, (f, fl) in kern.items()), flush=True)
