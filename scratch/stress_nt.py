"""Bitwise repeatability of gemm_nt under GPU sharing: a second process keeps the GPU busy while this one repeats the same
products and compares every output with the first one (gemm_nt has no atomics: any difference is a bug)."""
import os, sys, ctypes, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rlgym_ppo_amd import _native as N

L = N.lib(); dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
if len(sys.argv) > 1 and sys.argv[1] == "load":
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            b = a @ a
        torch.cuda.synchronize()
    sys.exit(0)

bg = subprocess.Popen([sys.executable, __file__, "load", "40"]) if os.environ.get("LOAD", "1") == "1" else None
torch.manual_seed(0)
shapes = [(512, 64, 128, 1), (512, 64, 128, 2), (512, 64, 64, 1), (512, 96, 64, 0), (512, 32, 64, 0), (512, 64, 96, 3), (512, 64, 64, 3), (512, 64, 32, 3),
          (4096, 256, 256, 1), (65536, 256, 256, 1)]
for variant in [int(v) for v in os.environ.get('VARIANTS', '0,1,2,3').split(',')]:
    N.check(L.rlppo_dbg_set(9, variant))
    for (M, n, k, epi) in shapes:
        A = torch.randn(M, k, device=dev); W = torch.randn(n, k, device=dev) * 0.1; bias = torch.randn(n, device=dev)
        mask = torch.randn(M, n, device=dev)
        C = torch.empty(M, n, device=dev)
        run = lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), k, None, P(W), k, P(bias), P(mask) if epi == 3 else None, n, P(C), n, M, n, k, epi))
        run(); ref = C.clone()
        bad = 0; worst = 0.0
        reps = 3000 if M <= 4096 else 200
        for r in range(reps):
            C.fill_(-1.0)
            run()
            if os.environ.get('SYNC'): torch.cuda.synchronize()
            if not torch.equal(C, ref):
                bad += 1
                worst = max(worst, (C - ref).abs().max().item())
        print(f"variant {variant} M={M} N={n} K={k} epi={epi}: {bad}/{reps} runs differ, worst abs diff {worst:.3e}", flush=True)
if bg: bg.wait()
