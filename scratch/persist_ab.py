"""gemm_nt persistent form (RLPPO_TUNE key 17) against the one-tile-per-workgroup form: bitwise equal outputs, launch times."""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
for M in (524288, 300077, 131072 + 5):
    A256 = torch.randn(M, 256, device="cuda"); Mk = torch.randn(M, 256, device="cuda"); A128 = torch.randn(M, 128, device="cuda"); A96 = torch.randn(M, 96, device="cuda")
    W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.randn(256, device="cuda")
    shapes = {"fwd hidden": (A256, 256, 256, 1), "fwd L0": (A128, 128, 256, 1), "fwd head": (A256, 256, 96, 0), "dX hidden": (A256, 256, 256, 3), "dX head": (A96, 96, 256, 3)}
    for name, (Am, K, n, epi) in shapes.items():
        outs, times = [], []
        for persist in (0, 1, 0, 1):
            N.check(L.rlppo_dbg_set(17, persist))
            C = torch.full((M, n), -7.0, device="cuda")
            f = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(Am), K, None, P(W), K, P(b) if epi != 3 else None, P(Mk) if epi == 3 else None, 256 if epi == 3 else 0, P(C), n, M, n, K, epi))
            f(); torch.cuda.synchronize()
            outs.append(C.clone())
            times.append(bench.time_region(f, 20, warm_s=0.15) * 1e3)
        same = torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])
        if M == 300077 and name == "fwd hidden":
            ref = torch.relu(A256.double() @ W.double().T + b.double())
            print("   max abs err vs fp64:", (outs[1].double() - ref).abs().max().item())
        print(f"M={M} {name:10s}: one-tile {min(times[0], times[2]):7.1f} us   persistent {min(times[1], times[3]):7.1f} us   bitwise equal: {same}", flush=True)
N.check(L.rlppo_dbg_set(17, 0))
