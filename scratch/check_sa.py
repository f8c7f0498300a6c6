import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlgym_ppo_amd import _native as N
L = N.lib(); dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
for variant in (1, 3):
    N.check(L.rlppo_dbg_set(9, variant))
    for (M, n, k, epi) in [(512, 64, 128, 1), (4096, 256, 256, 1), (512, 64, 128, 0)]:
        A = torch.randn(M, k, device=dev); W = torch.randn(n, k, device=dev) * 0.1; bias = torch.randn(n, device=dev)
        ref = A.double() @ W.double().T + bias.double()
        if epi == 1: ref = ref.clamp_min(0)
        outs = []
        for fillv in (-1.0, 7.0, -1.0):
            C = torch.full((M, n), fillv, device=dev)
            N.check(L.rlppo_dbg_gemm_nt(st(), P(A), k, None, P(W), k, P(bias), None, n, P(C), n, M, n, k, epi))
            torch.cuda.synchronize()
            outs.append(C.clone())
        err = [(o.double() - ref).abs().max().item() for o in outs]
        d01 = (outs[0] != outs[1]); d02 = (outs[0] != outs[2])
        print(f"variant {variant} M={M} N={n} K={k} epi={epi}: max err vs fp64 {err}; fill-dependent entries {int(d01.sum())}, run-to-run (same fill) {int(d02.sum())}")
        if d01.any():
            idx = torch.nonzero(d01)[:6].tolist()
            print("   first fill-dependent entries:", idx, [ (outs[0][i,j].item(), outs[1][i,j].item(), ref[i,j].item()) for i,j in idx[:3]])
