"""Data-parallel plumbing of the PPO update (the only place the hot path has a real exchange step).

The reference sums the gradients of the B/MB minibatch slices of a batch, each scaled by MB/B, before ONE
clip + Adam (rlgym_ppo/ppo/ppo_learner.py:134-193).  Dealing slice j to rank j % world and summing the flat
gradient arenas with one all-reduce is therefore the same computation; the permutation is drawn identically on
every rank (same seed), the buffer is replicated, clip + Adam run replicated on the reduced gradients.
One process per GPU; backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo" in the CPU tests.
"""
import torch


def dist_info():
    """(dist module or None, rank, world) for the default process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def slices_for_rank(n_slices, rank, world):
    """Minibatch slice numbers of one batch that `rank` computes: a contiguous block when the slices divide evenly over the
    ranks (so that a rank can evaluate its share in one fused pass over consecutive rows of the permutation), round-robin
    otherwise.  Either dealing gives the same gradient sum."""
    if n_slices % world == 0:
        per = n_slices // world
        return list(range(rank * per, (rank + 1) * per))
    return list(range(rank, n_slices, world))


def fuse_runs(slices, max_fused):
    """Consecutive slice numbers grouped into EQUAL runs of at most `max_fused` slices: [(first_slice, count), ...].
    Equal counts keep "mean over passes of the per-pass mean" identical to the reference's mean over minibatches."""
    n = len(slices)
    contiguous = all(slices[i] + 1 == slices[i + 1] for i in range(n - 1))
    k = 1
    if contiguous:
        k = max(d for d in range(1, max(1, min(max_fused, n)) + 1) if n % d == 0)
    return [(slices[i], k) for i in range(0, n, k)]


_direct = {"ready": False, "want": None}  # want: None = RLPPO_RCCL_DIRECT decides, True / False = set_allreduce_backend


def set_allreduce_backend(name):
    """A/B switch of the gradient exchange: "torch" = torch.distributed's all_reduce (default), "direct" = rlppo_allreduce
    (RCCL enqueued on the caller's stream by librlppo itself, include/rlppo.h).  Every rank must make the same choice.  The
    direct communicator is created on first use and kept; switching back and forth is free."""
    _direct["want"] = {"torch": False, "direct": True}[name]


def allreduce_backend():
    import os
    want = _direct["want"]
    if want is None:
        want = os.environ.get("RLPPO_RCCL_DIRECT") == "1"
    return "direct" if want else "torch"


def _direct_comm(dist):
    """True when the sum should go through rlppo_allreduce: the direct backend is selected and the default group runs on
    RCCL (with gloo -- the CPU tests, the one-GPU dry run -- the collective stays in torch.distributed).  The multi-process
    tests cover the torch.distributed route (gloo on the CPU; RCCL refuses two ranks on one GPU, so a one-GPU box can only
    exercise the one-rank communicator: tests/test_gpu_dp.py)."""
    if allreduce_backend() != "direct" or dist.get_backend() != "nccl":
        return False
    if not _direct["ready"]:
        import ctypes
        import os
        from . import _native as N
        L = N.lib()
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(rccl):  # the copy PyTorch already holds: one RCCL instance per process
            N.check(L.rlppo_comm_set_library(rccl.encode()))
        ident = ctypes.create_string_buffer(N.COMM_ID_BYTES)
        if dist.get_rank() == 0:
            N.check(L.rlppo_comm_unique_id(ident))
        box = [ident.raw]
        dist.broadcast_object_list(box, src=0)
        N.check(L.rlppo_comm_init(dist.get_rank(), dist.get_world_size(), ctypes.c_char_p(box[0])))
        _direct["ready"] = True
    return True


def all_reduce_sum(tensor, dist=None):
    """In-place sum over ranks of one flat buffer ([grad_policy | grad_value]: 1.37 MB for the 256x3 nets -- latency
    bound, so exactly one collective per optimiser step)."""
    if dist is None:
        dist, _, _ = dist_info()
    if dist is not None:
        if tensor.is_cuda and tensor.is_contiguous() and tensor.dtype in (torch.float32, torch.float64) and _direct_comm(dist):
            import ctypes
            from . import _native as N
            N.check(N.lib().rlppo_allreduce(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(tensor.data_ptr()),
                                            tensor.numel(), int(tensor.dtype == torch.float64)))
        else:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def run_virtual_ranks(learners, buffers):
    """N data-parallel ranks in ONE process on one device: drives PPOLearner.learn_steps of N replicas (identical construction,
    identical buffers) in lock step and performs every exchange as the sum, in rank order, of the tensors the replicas hand over
    -- what the all-reduce computes.  Rank r evaluates exactly the minibatch slices slices_for_rank deals it, so the partition of
    BASELINE configs[3] (8 ranks x one 65,536-row pass per optimiser step) runs literally on a one-GPU box, where the pool admits
    neither 8 processes on the card nor two RCCL ranks on one GPU.  Returns the N report dictionaries."""
    world = len(learners)
    gens = [l.learn_steps(b, r, world) for r, (l, b) in enumerate(zip(learners, buffers))]
    reports = [None] * world
    pending = [next(g) for g in gens]
    while True:
        total = pending[0].clone()
        for t in pending[1:]:
            total += t
        for t in pending:
            t.copy_(total)
        nxt, failed = [], None
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(None))
            except StopIteration as done:
                reports[r] = done.value
            except Exception as e:  # noqa: BLE001 -- a collective failure (OptimizerBarrierTimeout) is raised by every rank: let each recover
                failed = failed or e
        if failed is not None:
            raise failed
        if len(nxt) == 0:
            return reports
        if len(nxt) != world:
            raise RuntimeError("virtual ranks fell out of step: %d of %d reached the next exchange" % (len(nxt), world))
        pending = nxt
