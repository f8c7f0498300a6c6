import sys, ctypes, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
import bench
L = N.lib(); M = 65536; dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A128, A256, A96, A32 = (torch.randn(M, k, device=dev) for k in (128, 256, 96, 32))
idx = torch.randperm(M, device=dev); dW = torch.zeros(256*256, device=dev); db = torch.zeros(256, device=dev)
shapes = {"hidden256x256": (A256,256,A256,256,None,256,256), "L0_256x107_gather": (A256,256,A128,128,idx,256,107), "L0_256x107": (A256,256,A128,128,None,256,107), "head90x256": (A96,96,A256,256,None,90,256), "vhead1x256": (A32,32,A256,256,None,1,256)}
for variant in (2, 3):
  N.check(L.rlppo_dbg_set(10, variant)); print('tn variant', variant)
  for name,(dY,ny,X,kx,ri,out,in_) in shapes.items():
      res = []
      for rows in (128, 256, 384, 512, 768, 1024, 2048):
          N.check(L.rlppo_dbg_set(2, rows))
          fn = lambda: N.check(L.rlppo_dbg_gemm_tn(st(), P(dY), ny, ny, P(X), kx, P(ri), kx, P(dW), P(db), out, in_, M))
          res.append((rows, round(bench.time_region(fn, 20)*1e3, 1)))
      print(name, res)
