"""gemm_nt with three K tiles in flight (key 20) vs the two-buffer kernel: bitwise equality and time."""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
for M in (65536, 524288 + 3):
    A256 = torch.randn(M, 256, device="cuda"); Mk = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda")
    W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.randn(256, device="cuda")
    for name, (Am, K, epi) in {"fwd hidden": (A256, 256, 1), "fwd L0": (A128, 128, 1), "dX hidden": (A256, 256, 3)}.items():
        outs, times = [], []
        for v in (0, 1, 0, 1):
            N.check(L.rlppo_dbg_set(20, v))
            C = torch.full((M, 256), -7.0, device="cuda")
            f = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(Am), K, None, P(W), K, P(b) if epi != 3 else None, P(Mk) if epi == 3 else None, 256 if epi == 3 else 0, P(C), 256, M, 256, K, epi))
            f(); torch.cuda.synchronize(); outs.append(C.clone())
            times.append(bench.time_region(f, 20, warm_s=0.15) * 1e3)
        print(f"M={M} {name:10s}: 2 buffers {min(times[0], times[2]):7.1f} us   3 buffers {min(times[1], times[3]):7.1f} us   bitwise equal: {torch.equal(outs[0], outs[1])}", flush=True)
N.check(L.rlppo_dbg_set(20, 0))
