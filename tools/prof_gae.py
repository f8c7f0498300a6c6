"""BASELINE configs[2] GAE scan (8192 x 256 steps) launched REPS times on preallocated outputs: the target of
`rocprofv3 --kernel-trace --stats` and of the FETCH_SIZE / WRITE_SIZE PMC passes (tools/round_profile.sh)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
L = N.lib()
rs = np.random.RandomState(0)
n_seg, seg = 8192, 256
n = n_seg * seg
rews, values = rs.randn(n).astype(np.float32), rs.randn(n + 1).astype(np.float32)
dones = (rs.rand(n) < 0.005).astype(np.float32)
trunc = np.zeros(n, np.float32)
ends = np.arange(seg - 1, n, seg)
is_done = rs.rand(n_seg) < 0.5
dones[ends[is_done]] = 1
dones[ends[~is_done]] = 0
trunc[ends[~is_done]] = 1
d = lambda x: torch.as_tensor(x).cuda()
R, D, T, V = d(rews), d(dones), d(trunc), d(values)
vt, adv, ret = (torch.empty(n, device="cuda") for _ in range(3))
ws = torch.zeros(int(L.rlppo_gae_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(int(os.environ.get("REPS", 120))):
    N.check(L.rlppo_gae(st, P(R), P(D), P(T), P(V), n, 0.99, 0.95, float(np.float32(1.7)), P(vt), P(adv), P(ret), P(ws), ws.numel()))
torch.cuda.synchronize()
