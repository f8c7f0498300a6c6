"""Host cost of the bit-exact rollout noise (rlppo_torch_cpu_exponential) on this box: per thread count, for the 4096 x 90 draw
of one configs[1] rollout step, next to torch's own exponential_."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
L = N.lib()
n = 4096 * 90
a = torch.get_rng_state().numpy().copy()
out = np.empty(n, np.float32)
for th in (1, 2, 4, 8, 16, 32):
    call = lambda: N.check(L.rlppo_torch_cpu_exponential(ctypes.c_void_p(a.ctypes.data), a.size, n, 1.0, ctypes.c_void_p(out.ctypes.data), th))
    for _ in range(3):
        call()
    t = time.perf_counter()
    for _ in range(20):
        call()
    print(f"threads {th:3d}: {(time.perf_counter() - t) / 20 * 1e3:7.3f} ms")
x = torch.empty(4096, 90)
t = time.perf_counter()
for _ in range(10):
    x.exponential_(1)
print(f"torch exponential_: {(time.perf_counter() - t) / 10 * 1e3:7.3f} ms   (host threads: {os.cpu_count()})")
