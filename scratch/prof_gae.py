import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rlgym_ppo_amd.util import torch_functions
rs = np.random.RandomState(0); n = 8192*256
d = lambda x: torch.as_tensor(x).cuda()
R, V = d(rs.randn(n).astype(np.float32)), d(rs.randn(n+1).astype(np.float32))
D = d((rs.rand(n) < 0.005).astype(np.float32)); T = torch.zeros(n, device="cuda"); T[255::256] = 1
for _ in range(30): torch_functions.gae_device(R, D, T, V, 0.99, 0.95, 1.7)
torch.cuda.synchronize()
