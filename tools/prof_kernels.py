"""Runs every kernel flavour of one pass of the cfg2 update (M rows) + the GAE scan a few times (for rocprofv3 --pmc /
--kernel-trace)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N  # noqa: E402
from rlgym_ppo_amd.util import torch_functions  # noqa: E402

L = N.lib()
M = int(os.environ.get("M", 524288))  # rows per launch: one GPU evaluates the 8 minibatches of a batch in one pass
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
dev = "cuda"
K0 = int(L.rlppo_padded_width(107))  # 112: the first layer's padded contraction
A128, A256, A96, A32 = (torch.randn(M, k, device=dev) for k in (K0, 256, 96, 32))
A256b = torch.randn(M, 256, device=dev)  # second operand of dW / mask of dX: distinct memory, as in the update
W = torch.randn(256, 256, device=dev) * 0.05
bias = torch.zeros(256, device=dev)
C256, C96, C32 = (torch.empty(M, k, device=dev) for k in (256, 96, 32))
idx = torch.randperm(M, device=dev)
dW = torch.zeros(256 * 256, device=dev)
db = torch.zeros(256, device=dev)
tn_ws = torch.empty(max(int(L.rlppo_dbg_gemm_tn_workspace_bytes(o, i, M)) for o, i in ((256, 256), (256, 107), (90, 256))),
                    dtype=torch.uint8, device=dev)
bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, 256)), dtype=torch.uint8, device=dev)  # ReLU bitmask: forward writes, dX reads
# [r5] the grouped weight-gradient launch of a pass (7 products: 2 first-layer, 4 hidden, the policy head) + its one reduction
gshapes = [(256, 107, A256, 256, A128, K0), (256, 256, A256, 256, A256b, 256), (256, 256, A256b, 256, A256, 256), (90, 256, A96, 96, A256, 256),
           (256, 107, A256b, 256, A128, K0), (256, 256, A256, 256, A256b, 256), (256, 256, A256b, 256, A256, 256)]
gprods = (N.TnProduct * len(gshapes))()
gkeep = []
for q, (out, in_, dY, ny, X, kx) in zip(gprods, gshapes):
    gw, gb = torch.zeros(out * in_, device=dev), torch.zeros(out, device=dev)
    gkeep.append((gw, gb))
    q.dY, q.ldy, q.ny_valid, q.X, q.ldx, q.kx_valid = dY.data_ptr(), ny, ny, X.data_ptr(), kx, kx
    q.dW, q.db, q.out, q.in_, q.rowtab, q.src_rows = gw.data_ptr(), gb.data_ptr(), out, in_, None, 0
g_ws = torch.empty(int(L.rlppo_dbg_gemm_tn_group_workspace_bytes(gprods, len(gshapes), M)), dtype=torch.uint8, device=dev)
reps = int(os.environ.get("REPS", 3))
for _ in range(reps):
    N.check(L.rlppo_dbg_gemm_tn_group(st(), gprods, len(gshapes), M, P(g_ws), g_ws.numel()))
    N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A256), 256, P(W), 256, P(bias), P(C256), 256, M, 256, 256, 1, P(bits)))
    N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A128), K0, P(W), K0, P(bias), P(C256), 256, M, 256, K0, 1, P(bits)))
    N.check(L.rlppo_dbg_gemm_nt(st(), P(A256), 256, P(W), 256, P(bias), None, 0, P(C96), 96, M, 96, 256, 0))
    N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A256), 256, P(W), 256, None, P(C256), 256, M, 256, 256, 3, P(bits)))
    N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A96), 96, P(W), 96, None, P(C256), 256, M, 256, 96, 3, P(bits)))
    N.check(L.rlppo_dbg_gemm_tn(st(), P(A256), 256, 256, P(A256b), 256, 256, P(dW), P(db), 256, 256, M, P(tn_ws), tn_ws.numel()))
    N.check(L.rlppo_dbg_gemm_tn(st(), P(A256), 256, 256, P(A128), K0, K0, P(dW), P(db), 256, 107, M, P(tn_ws), tn_ws.numel()))
    N.check(L.rlppo_dbg_gemm_tn(st(), P(A96), 96, 96, P(A256), 256, 256, P(dW), P(db), 90, 256, M, P(tn_ws), tn_ws.numel()))
rs = np.random.RandomState(0)
n = 8192 * 256
d = lambda x: torch.as_tensor(x).cuda()
R, V = d(rs.randn(n).astype(np.float32)), d(rs.randn(n + 1).astype(np.float32))
D = d((rs.rand(n) < 0.005).astype(np.float32))
T = torch.zeros(n, device="cuda")
T[255::256] = 1
torch.cuda.synchronize()  # (the GAE scan has its own script: tools/prof_gae.py)
