"""What would one grouped launch (both nets' same-shaped layer) buy over two launches on one stream / on two streams?"""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from rlgym_ppo_amd import _native as N
L = N.lib(); M = 65536
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
A2 = torch.randn(2 * M, 256, device="cuda"); W = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda"); C2 = torch.empty(2 * M, 256, device="cuda")
s0, s1 = torch.cuda.current_stream(), torch.cuda.Stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
def launch(stream, off, rows, epi=1, mask=None):
    N.check(L.rlppo_dbg_gemm_nt(sp(stream), ctypes.c_void_p(A2.data_ptr() + off * 1024), 256, None, P(W), 256, P(b) if epi != 3 else None,
                                ctypes.c_void_p(A2.data_ptr() + off * 1024) if epi == 3 else None, 256, ctypes.c_void_p(C2.data_ptr() + off * 1024), 256, rows, 256, 256, epi))
def two_seq(): launch(s0, 0, M); launch(s0, M, M)
def one_big(): launch(s0, 0, 2 * M)
ev = torch.cuda.Event()
def two_streams():
    ev.record(s0); s1.wait_event(ev)
    launch(s0, 0, M); launch(s1, M, M)
    e2 = torch.cuda.Event(); e2.record(s1); s0.wait_event(e2)
for name, fn in (("two launches, one stream", two_seq), ("one launch of 2x the rows", one_big), ("two launches, two streams", two_streams)):
    print("%-28s %.1f us per pair" % (name, bench.time_region(fn, 20, warm_s=0.3) * 1e3))
