"""Does a wave that streams MFMAs slow the instruction issue of the other wave on its SIMD?  (tools; GPU only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rlgym_ppo_amd import _native as N
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag as D

L = N.lib()
dev = torch.device("cuda:0")
buf = torch.randn(1 << 23, device=dev)
out = torch.empty(512 * 256, device=dev)
cyc = torch.zeros(1024 * 2, dtype=torch.int64, device=dev)
P = lambda t: t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream
iters = 2048
for flags, label in ((1, "MFMA+LDS | 8 loads, no VALU"), (3, "MFMA+LDS | 8 loads + 64-bit address VALU"),
                     (16 + 1, "MFMA+LDS | 16 v_add_u32"), (32 + 1, "MFMA+LDS | 16 v_lshl_add_u64"),
                     (64 + 1, "MFMA+LDS | 16 v_pk_add_f32"), (128 + 1, "MFMA+LDS | 16 v_fma_f32"),
                     (16, "MFMA only | 16 v_add_u32"), (32, "MFMA only | 16 v_lshl_add_u64"),
                     (16 + 4, "idle | 16 v_add_u32"), (32 + 4, "idle | 16 v_lshl_add_u64"), (64 + 4, "idle | 16 v_pk_add_f32")):
    fn = lambda: D.check(D.DL.rlppo_dbg_probe_coissue(st(), P(buf), flags, iters, P(cyc), P(out)))
    ms = bench.time_region(fn, 3)
    c = cyc.cpu().numpy().reshape(-1, 2).astype(np.float64)
    nb = iters // 4
    mf = 256 * 4 * iters * 64 * 2048 / (ms * 1e-3) / 1e12  # 64 MFMAs x 2048 flop per iteration per wave (half the waves)
    print(f"flags {flags:2d} {label:44s}: kernel {ms*1e3:7.1f} us, streamer MFMA rate {mf:6.1f} TF (of 78.6 for half the waves) | "
          f"loader: {c[:,0].mean()/nb:7.0f} cycles for the timed block, {c[:,1].mean()/nb:7.0f} per batch overall", flush=True)
