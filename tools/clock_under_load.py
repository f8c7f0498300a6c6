"""Shader clock (s_memtime ticks per 100 MHz reference tick) inside the stamped gemm_nt launch: cold, and after sustained load."""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rlgym_ppo_amd import _native as N
L = N.lib(); M = 65536
P = lambda t: ctypes.c_void_p(t.data_ptr()); st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda")
C = torch.empty(M, 256, device="cuda"); nwg = (M // 128) * 2
stamps = torch.zeros(nwg * 4 * 10, dtype=torch.int64, device="cuda")
def clock():
    N.check(L.rlppo_dbg_gemm_nt_stamped(st(), P(A), 256, P(W), 256, P(b), P(C), 256, M, 256, 256, P(stamps), 64))
    raw = stamps.cpu().numpy(); ab = raw[nwg * 32:].reshape(nwg, 4, 2).astype(np.float64)
    return ab[:, :, 0].sum() / ab[:, :, 1].sum() * 100
gemm = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, None, P(W), 256, P(b), None, 0, P(C), 256, M, 256, 256, 1))
torch.cuda.synchronize(); time.sleep(2.0)
print(f"cold: {clock():.0f} MHz, hidden fwd {bench.time_region(gemm, 10)*1e3:.1f} us")
for secs in (0.5, 1.0, 2.0, 4.0):
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(50): gemm()
        torch.cuda.synchronize()
    print(f"after {secs:.1f} s more of back-to-back GEMMs: {clock():.0f} MHz, hidden fwd {bench.time_region(gemm, 10)*1e3:.1f} us")
# clock while ANOTHER stream keeps the chip busy with a different GEMM flavour (the two-stream update's situation)
side = torch.cuda.Stream()
X2 = torch.randn(M, 256, device="cuda"); dW = torch.zeros(256 * 256, device="cuda"); db = torch.zeros(256, device="cuda")
for _ in range(3):
    with torch.cuda.stream(side):
        for _ in range(300):
            N.check(L.rlppo_dbg_gemm_tn(ctypes.c_void_p(side.cuda_stream), P(X2), 256, 256, P(A), 256, None, 256, P(dW), P(db), 256, 256, M))
    for _ in range(100): gemm()
    c = clock()
    torch.cuda.synchronize()
    print(f"with dW GEMMs running on a second stream: {c:.0f} MHz")
