"""How long does the HOST need to enqueue one learn() step, and how does that compare with the GPU time?  Emulates the
per-rank share of an N-rank run on one GPU (no process group, no all-reduce): a rank of an N-rank job enqueues the same
per-epoch host work but only 8/N of the minibatch slices, so `learn()` wall time here = the floor of an N-rank step."""
import ctypes, os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rlgym_ppo_amd import _native as N
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
learner.n_epochs = 10
# the two phases of the permutation on this host
L = N.lib()
st = np.empty(625, np.uint32); L.rlppo_mt19937_seed(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1)
n = 524288; tg = np.empty(n + 8, np.uint32); out = np.empty(n, np.int64)
for name, fn in (("fused permutation", lambda: L.rlppo_mt19937_permutation(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, ctypes.c_void_p(out.ctypes.data))),
                 ("draw_targets", lambda: L.rlppo_mt19937_draw_targets(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, ctypes.c_void_p(tg.ctypes.data))),
                 ("apply_swap_targets", lambda: L.rlppo_apply_swap_targets(n, ctypes.c_void_p(tg.ctypes.data), ctypes.c_void_p(out.ctypes.data)))):
    fn(); t0 = time.perf_counter()
    for _ in range(10): fn()
    print(f"{name}: {(time.perf_counter() - t0) * 100:.3f} ms per 524,288 indices", flush=True)
learner.learn(buf); torch.cuda.synchronize()
import rlgym_ppo_amd.ppo.ppo_learner as PL
orig = PL.slices_for_rank
for world_emul in (1, 2, 4, 8):
    PL.slices_for_rank = (lambda n, r, w, we=world_emul: list(range(0, n // we)))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        learner.learn(buf)
        t_host = time.perf_counter() - t0          # includes the final stats .cpu() sync of learn()
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    PL.slices_for_rank = orig
    print(f"minibatch share 1/{world_emul}: learn() (10 epochs) {t_all*1e3:.1f} ms -> floor of a {world_emul}-rank step "
          f"= {5242880 / t_all / 1e6:.1f} M samples/s", flush=True)
