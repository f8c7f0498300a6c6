"""fp32 MFMA ceiling on this box: register-only v_mfma_f32_16x16x4_f32 loop, 1-3 workgroups per CU."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag as D
import bench
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
iters = 4000
for blocks in (256, 512, 768, 1024):
    out = torch.empty(blocks * 256, device="cuda"); clk = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda")
    fn = lambda: D.check(D.DL.rlppo_dbg_mfma_probe(st(), ctypes.c_void_p(out.data_ptr()), blocks, iters, ctypes.c_void_p(clk.data_ptr())))
    ms = bench.time_region(fn, 5)
    flop = blocks * 4 * iters * 64 * 2048
    c = clk.cpu().numpy().reshape(-1, 2).astype(np.float64)
    ghz = np.median(c[:, 0] / c[:, 1]) * 0.1
    print(f"blocks {blocks:5d}: {ms:8.3f} ms  {flop/ms/1e9:7.1f} TFLOP/s  in-kernel clock {ghz:.2f} GHz  cycles/MFMA/SIMD {np.median(c[:,0])/(iters*64)/ (max(1, blocks//256)):.1f}")
