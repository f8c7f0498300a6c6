"""Per-CU vector-memory throughput by lane->address pattern and by where the data lives (tools; GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rlgym_ppo_amd import _native as N
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag as D

L = N.lib()
dev = torch.device("cuda:0")
buf = torch.randn(1 << 28, device=dev)  # 1 GiB
P = lambda t: t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream
names = {0: "8 rows x 128 B", 1: "1 KB contiguous", 2: "4 rows x 256 B", 3: "16 x 64 B"}
for wg_per_cu in (2, 4, 8):
    blocks = 256 * wg_per_cu
    out = torch.empty(blocks * 256, device=dev)
    for span, where in ((1 << 30, "HBM 1 GiB"), (1 << 27, "MALL 128 MiB"), (1 << 24, "L2 16 MiB (2 MiB/XCD)")):
        nwaves = blocks * 4
        iters = max(64, min(4096, (1 << 32) // (nwaves * 8192)))
        row = []
        for pat in range(4):
            fn = lambda: D.check(D.DL.rlppo_dbg_probe_ld(st(), pat, blocks, P(buf), span, iters, P(out)))
            ms = bench.time_region(fn, 5)
            byts = nwaves * 8192 * iters
            row.append(f"{names[pat]}: {byts/ms/1e9:6.2f} TB/s = {byts/(ms*1e-3)/2.4e9/256:5.1f} B/clk/CU")
        print(f"{wg_per_cu*4:2d} waves/CU, {where:24s} | " + " | ".join(row), flush=True)
