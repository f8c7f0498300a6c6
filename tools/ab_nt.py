"""A/B of gemm_nt: per-lane 64-bit pointers (gemm.hip) vs scalar-addressed buffer loads (gemm_sa.hip), interleaved."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536; dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A128, A256, A96, A32 = (torch.randn(M, k, device=dev) for k in (128, 256, 96, 32))
W = torch.randn(256, 256, device=dev) * 0.05; bias = torch.zeros(256, device=dev)
C256, C96, C32 = (torch.empty(M, k, device=dev) for k in (256, 96, 32)); idx = torch.randperm(M, device=dev)
shapes = {
 "fwd L0 gather 128->256": (A128,128,idx,128,C256,256,256,128,1,None),
 "fwd hidden 256->256": (A256,256,None,256,C256,256,256,256,1,None),
 "fwd head 256->96": (A256,256,None,256,C96,96,96,256,0,None),
 "fwd vhead 256->32": (A256,256,None,256,C32,32,32,256,0,None),
 "dX hidden mask": (A256,256,None,256,C256,256,256,256,3,A256),
 "dX head 96->256": (A96,96,None,96,C256,256,256,96,3,A256),
 "dX vhead 32->256": (A32,32,None,32,C256,256,256,32,3,A256)}
variants = {0: "vector-addr", 1: "scalar-addr", 2: "LDS-DMA BK32", 3: "LDS-DMA BK16"}
for name,(A,lda,ri,ldb,C,ldc,n,k,epi,mask) in shapes.items():
    fn = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), lda, P(ri), P(W), ldb, P(bias), P(mask), n, P(C), ldc, M, n, k, epi))
    t = {v: [] for v in variants}
    for _ in range(5):
        for v in variants:
            N.check(L.rlppo_dbg_set(9, v)); t[v].append(bench.time_region(fn, 10))
    N.check(L.rlppo_dbg_set(9, 3))
    fl = 2*M*n*k
    print("%-24s" % name + " | ".join("%s %6.1f us %6.1f TF" % (variants[v], np.median(t[v])*1e3, fl/np.median(t[v])/1e9) for v in variants))
