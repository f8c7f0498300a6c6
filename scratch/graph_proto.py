"""Can rlppo_ppo_minibatch (two internal streams, events) be captured into a HIP graph through torch.cuda.graph, and
what does replay cost against direct launches?"""
import os, sys, time, ctypes, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rlgym_ppo_amd import _native as N
from rlgym_ppo_amd.engine import stream_ptr
with contextlib.redirect_stdout(sys.stderr):
    learner, buf = bench.build_workload("cuda:0")
L = N.lib()
learner.learn(buf); torch.cuda.synchronize()
args = learner._minibatch_args(buf)
MB = learner.mini_batch_size
idx = torch.randperm(524288, device="cuda")
pa, va = learner.policy.arena, learner.value_net.arena
pa.ensure_packed(); va.ensure_packed()
def direct(j):
    args.slot = 0; args.workspace = learner._slot_ws[0]
    args.idx = idx.data_ptr() + 8 * j * MB; args.mb = MB
    N.check(L.rlppo_ppo_minibatch(stream_ptr(), ctypes.byref(args)))
learner._grad_all.zero_()
for j in range(8): direct(j)
torch.cuda.synchronize()
g_direct = learner._grad_all.clone()
# host time and GPU time of 8 direct minibatches
torch.cuda.synchronize(); t0 = time.perf_counter()
for j in range(8): direct(j)
t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"direct: host enqueue {t_host*1e3:.2f} ms, total {t_all*1e3:.2f} ms for 8 minibatches")
# capture
s = torch.cuda.Stream()
graphs = []
with torch.cuda.stream(s):
    for j in range(8):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            direct(j)
        graphs.append(g)
torch.cuda.synchronize()
learner._grad_all.zero_()
for g in graphs: g.replay()
torch.cuda.synchronize()
g_graph = learner._grad_all.clone()
print("graph vs direct grads: max abs diff", (g_graph - g_direct).abs().max().item(), "rel", ((g_graph - g_direct).abs().max() / g_direct.abs().max()).item())
torch.cuda.synchronize(); t0 = time.perf_counter()
for g in graphs: g.replay()
t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"graph replay: host enqueue {t_host*1e3:.2f} ms, total {t_all*1e3:.2f} ms for 8 minibatches")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        for g in graphs: g.replay()
    torch.cuda.synchronize(); print(f"  graph: {(time.perf_counter()-t0)/40*1e3:.3f} ms per minibatch")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        for j in range(8): direct(j)
    torch.cuda.synchronize(); print(f"  direct: {(time.perf_counter()-t0)/40*1e3:.3f} ms per minibatch")
