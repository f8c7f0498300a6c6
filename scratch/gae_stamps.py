import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
from rlgym_ppo_amd.util import torch_functions
L = N.lib()
raw = ctypes.CDLL(N.LIB_PATH)
rs = np.random.RandomState(0); n = 8192 * 256
d = lambda x: torch.as_tensor(x).cuda()
R, V = d(rs.randn(n).astype(np.float32)), d(rs.randn(n + 1).astype(np.float32))
D = d((rs.rand(n) < 0.005).astype(np.float32)); T = torch.zeros(n, device="cuda"); T[255::256] = 1
fn = lambda: torch_functions.gae_device(R, D, T, V, 0.99, 0.95, 1.7)
bench.time_region(fn, 1, warm_s=0.3)
for dma in (0, 1):
    N.check(L.rlppo_dbg_set(17, dma))
    for _ in range(5): fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    assert raw.rlppo_dbg_gae_stamps_copy(buf, 1024 * 8) == 0
    a = np.array(buf, dtype=np.float64).reshape(1024, 8)
    m = a.mean(0)
    print(f"dma={dma}: per workgroup cycles: loads+local composite {m[0]:.0f} | scan+barrier {m[1]:.0f} | look-ahead/carry+barrier {m[2]:.0f} | outputs+stores issued {m[3]:.0f} | total {m[4]:.0f}  (kernel {bench.time_region(fn, 20)*1e3:.1f} us)")
