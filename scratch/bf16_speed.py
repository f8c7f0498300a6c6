import os, sys, ctypes, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rlgym_ppo_amd import _native as N
L = N.lib(); M = 65536
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for (n, k) in ((256, 256), (512, 512), (512, 256)):
    A = torch.randn(M, k, device="cuda"); W = torch.randn(n, k, device="cuda") * 0.05; b = torch.zeros(n, device="cuda"); C = torch.empty(M, n, device="cuda")
    fn = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), k, None, P(W), k, P(b), None, 0, P(C), n, M, n, k, 1))
    t32 = bench.time_region(fn, 20, warm_s=0.2)
    print(f"M={M} N={n} K={k}: fp32 {t32*1e3:.1f} us ({2*M*n*k/t32/1e9:.0f} TF)")
from rlgym_ppo_amd.engine import set_inference_precision
from rlgym_ppo_amd.ppo import ValueEstimator
val = ValueEstimator(231, (512, 512, 512, 512), "cuda:0")
obs = torch.randn(262144, 231).numpy()
for mode in ("fp32", "bf16", "fp32", "bf16"):
    set_inference_precision(mode)
    x = torch.as_tensor(obs).cuda()
    fn = lambda: val(x)
    try:
        t = bench.time_region(fn, 5, warm_s=0.3)
        print(f"value pass 262,144 x 231 -> 512x4 -> 1, {mode}: {t:.2f} ms")
    except Exception as e:
        print("value pass failed:", repr(e)[:200]); break
set_inference_precision("fp32")
