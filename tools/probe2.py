"""Round-2 issue-stall probe: a synthetic MFMA + LDS-read loop of the diagnostic library (csrc/diag, rlppo_dbg_probe2) timed by mode / threads /
blocks -- what a K step of the fp32 NT kernel costs when nothing but issue limits it.  usage: python tools/probe2.py (needs the diag library:
make -C rlgym_ppo_amd/csrc diag)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlgym_ppo_amd import _native as N
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _diag as D
import bench
L = N.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
W = torch.randn(256, 256, device="cuda"); chunks = 4096
names = {0: "MFMA only", 1: "+A frags LDS (2)", 2: "+B frags LDS (8)", 3: "+A+B LDS (10)", 4: "+B frags global (8)", 5: "+A LDS +B global"}
for threads, blocks in ((256, 256), (256, 512), (512, 256)):
    out = torch.empty(blocks * threads, device="cuda")
    for mode in (0, 1, 2, 3, 4, 5):
        fn = lambda: D.check(D.DL.rlppo_dbg_probe2(st(), mode, threads, blocks, ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(out.data_ptr()), chunks))
        ms = bench.time_region(fn, 3)
        flop = blocks * (threads // 64) * chunks * 64 * 2048
        print(f"threads {threads} blocks {blocks} {names[mode]:22s}: {flop/ms/1e9:7.1f} TFLOP/s")
