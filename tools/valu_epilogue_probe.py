"""How much does a vector-ALU-heavy epilogue cost a fp32 MFMA GEMM at the policy head's shape?  The same launch (524,288 x 96 x 256,
gemm_nt_dma_kernel<6, ., 16>) with the plain bias epilogue and with the tanh epilogue (tanhf on every output: ~30-40 VALU instructions
per element, 48 elements per lane -- about the instruction count a loss epilogue would add).  HIP events, interleaved.
usage: python tools/valu_epilogue_probe.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlgym_ppo_amd import _native as N

L = N.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 524288
A = torch.randn(M, 256, device="cuda") * 0.1
W = torch.randn(96, 256, device="cuda") * 0.05
b = torch.zeros(96, device="cuda")
C = torch.empty(M, 96, device="cuda")
res = {0: [], 2: []}
for _ in range(3):
    for epi in (0, 2):
        res[epi].append(bench.time_region(lambda: N.check(L.rlppo_dbg_gemm_nt(st(), P(A), 256, P(W), 256, P(b), None, 0, P(C), 96, M, 96, 256, epi)), 20, warm=5))
print("policy-head forward 524,288 x 96 x 256: bias epilogue %.4f ms, tanh epilogue %.4f ms" % (np.median(res[0]), np.median(res[2])))
