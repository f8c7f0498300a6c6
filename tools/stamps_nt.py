"""Round-2 clock stamps inside the fp32 NT kernel (diagnostic library, rlppo_dbg_gemm_nt_stamped): where a workgroup's time goes between its
first DMA issue and its last store.  usage: python tools/stamps_nt.py (needs the diag library: make -C rlgym_ppo_amd/csrc diag)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag as D
L = N.lib(); M = 65536
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
A = torch.randn(M, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.randn(256, device="cuda")
C = torch.empty(M, 256, device="cuda"); nwg = (M // 128) * 2
stamps = torch.zeros(nwg * 4 * 10, dtype=torch.int64, device="cuda")
import bench
names = ["prologue", "issue loads", "frags+MFMA", "wait vmcnt", "LDS writes", "barrier", "epilogue", "TOTAL"]
for mode, label in ((0, "real"), (1, "A rows from L2"), (2, "no stores"), (64, "scalar-addressed kernel")):
    fn = lambda: D.check(D.DL.rlppo_dbg_gemm_nt_stamped(st(), P(A), 256, P(W), 256, P(b), P(C), 256, M, 256, 256, P(stamps), mode))
    ms = bench.time_region(fn, 5)
    raw = stamps.cpu().numpy()
    s = raw[:nwg * 32].reshape(nwg, 4, 8).astype(np.float64).mean(axis=(0, 1))
    print(f"mode {mode} {label:22s}: {ms*1e3:6.1f} us | " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, s)))
    if mode & 64:
        ab = raw[nwg * 32:].reshape(nwg, 4, 2).astype(np.float64)
        print(f"   s_memtime ticks per 100 MHz reference tick: {ab[:,:,0].sum()/ab[:,:,1].sum():.2f} -> s_memtime counts at "
              f"{ab[:,:,0].sum()/ab[:,:,1].sum()*100:.0f} MHz; mean wave lifetime {ab[:,:,1].mean()/100:.1f} us")
