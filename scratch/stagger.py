import os, sys, ctypes, torch
sys.path.insert(0, ".")
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A = torch.randn(M, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda"); C = torch.empty(M, 256, device="cuda")
fn = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, None, P(W), 256, P(b), None, 0, P(C), 256, M, 256, 256, 1))
print("RLPPO_STAGGER", os.environ.get("RLPPO_STAGGER"), "fwd hidden %.1f us" % (bench.time_region(fn, 30, warm_s=0.3) * 1e3))
