"""ReLU bitmask (key 19) A/B: isolated launches and the whole update."""
import os, sys, ctypes, time, contextlib, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for M in (65536, 524288):
    A = torch.randn(M, 256, device="cuda"); H = torch.relu(torch.randn(M, 256, device="cuda")); W = torch.randn(256, 256, device="cuda") * 0.05
    b = torch.zeros(256, device="cuda"); C = torch.empty(M, 256, device="cuda")
    bits = torch.zeros(int(L.rlppo_dbg_gemm_nt_bits_bytes(M, 256)), dtype=torch.uint8, device="cuda")
    f = {"fwd plain": lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, None, P(W), 256, P(b), None, 0, P(C), 256, M, 256, 256, 1)),
         "fwd +bits": lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), 256, P(W), 256, P(b), P(C), 256, M, 256, 256, 1, P(bits))),
         "dX mask=h": lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, None, P(W), 256, None, P(H), 256, P(C), 256, M, 256, 256, 3)),
         "dX bits": lambda: N.check(L.rlppo_dbg_gemm_nt_bits(st(), P(A), 256, P(W), 256, None, P(C), 256, M, 256, 256, 3, P(bits)))}
    res = {k: min(bench.time_region(fn, 20, warm_s=0.15) for _ in range(2)) * 1e3 for k, fn in f.items()}
    print(f"M={M}: " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
for rep in range(2):
    for v in (0, 1):
        N.check(L.rlppo_dbg_set(19, v))
        learner.learn(buf); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): learner.learn(buf)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"update, bitmask {v}: {dt*1e3:.2f} ms per 10-epoch learn() = {5242880/dt/1e6:.2f} M samples/s", flush=True)
