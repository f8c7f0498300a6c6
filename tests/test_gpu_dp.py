"""Data-parallel PPO update on the GPU with the real kernels: two ranks (gloo backend, both on cuda:0 -- the test box has
one GPU; on the 8-GPU node the same code runs one rank per GPU over RCCL) must reproduce the single-process update.
Slices are dealt round-robin, the flat [grad_policy | grad_value] arena is all-reduced once per optimiser step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed=7, batch=2048, hidden=(64, 64)):
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    torch.manual_seed(seed)
    learner = PPOLearner(107, 90, 0, hidden, hidden, (0.1, 1.0), batch, 2, 3e-4, 3e-4, 0.2, 0.005, 512, "cuda:0")
    rs = np.random.RandomState(seed)
    n = 4096
    obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32)
    noise = torch.as_tensor(rs.exponential(size=(n, 90)).astype(np.float32))
    act, logp = learner.policy.get_action(obs, noise=noise)
    buf = ExperienceBuffer(n, seed, "cpu")
    z = np.zeros(n, np.float32)
    buf.submit_experience(obs, act.numpy().astype(np.float32), logp.numpy() + 0.1 * rs.randn(n).astype(np.float32), z, obs, z, z,
                          rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
    return learner, buf


def _worker(rank, world, port, out, batch=2048, precision="fp32", hidden=(64, 64)):
    sys.path.insert(0, ROOT)
    import contextlib
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        learner, buf = _build(batch=batch, hidden=hidden)
    if precision != "fp32":
        from rlgym_ppo_amd.engine import set_update_precision
        set_update_precision(precision)
    report = learner.learn(buf)
    out[rank] = (learner.policy.arena.flat.cpu(), learner.value_net.arena.flat.cpu(), report)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    learner, buf = _build()
    ref_report = learner.learn(buf)
    ref_p, ref_v = learner.policy.arena.flat.cpu(), learner.value_net.arena.flat.cpu()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        p, v, report = out[rank]
        # summation order differs (atomics + all-reduce); Adam turns 1e-7 gradient noise into up to ~1e-5 relative parameter
        # noise after 4 steps (two identical single-process runs already differ by up to 9e-6: scratch/determinism.py), while
        # a real defect shows up as O(lr / max|p|) ~ 2e-3
        assert ((p - ref_p).abs().max() / ref_p.abs().max()).item() < 5e-5
        assert ((v - ref_v).abs().max() / ref_v.abs().max()).item() < 5e-5
        for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction",
                  "Policy Update Magnitude", "Value Function Update Magnitude"):
            assert abs(report[k] - ref_report[k]) <= 2e-5 * max(abs(ref_report[k]), 1e-3) + 1e-7, (k, report[k], ref_report[k])
        assert report["Cumulative Model Updates"] == ref_report["Cumulative Model Updates"] == 4
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])  # replicas stay bit-identical


def test_two_ranks_equal_one_rank_in_the_bf16_update_precision():
    """The same check in the bf16 update precision (256-wide nets: the bf16 forward / dX / dW kernels and the critic's narrow-head
    kernels): per-row arithmetic does not depend on which rank owns a slice and weight gradients are never rounded, so two ranks
    reproduce one rank up to the summation order of the all-reduce."""
    from rlgym_ppo_amd.engine import set_update_precision
    learner, buf = _build(hidden=(256, 256))
    set_update_precision("bf16")
    try:
        ref_report = learner.learn(buf)
    finally:
        set_update_precision("fp32")
    ref_p, ref_v = learner.policy.arena.flat.cpu(), learner.value_net.arena.flat.cpu()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, 2048, "bf16", (256, 256)), nprocs=2, join=True)
    for rank in (0, 1):
        p, v, report = out[rank]
        assert ((p - ref_p).abs().max() / ref_p.abs().max()).item() < 5e-5
        assert ((v - ref_v).abs().max() / ref_v.abs().max()).item() < 5e-5
        for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
            assert abs(report[k] - ref_report[k]) <= 2e-5 * max(abs(ref_report[k]), 1e-3) + 1e-7, (k, report[k], ref_report[k])
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_uneven_slices_report_the_mean_over_all_passes():
    """3 minibatch slices on 2 ranks (round-robin: rank 0 runs two passes, rank 1 one): the pass count travels with the
    all-reduced statistics, so both ranks report the single-process means (the round-1 code divided by 4 on rank 0 and by
    2 on rank 1) and the parameters still match."""
    learner, buf = _build(batch=1536)
    ref_report = learner.learn(buf)
    ref_p = learner.policy.arena.flat.cpu()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, 1536), nprocs=2, join=True)
    for rank in (0, 1):
        p, v, report = out[rank]
        assert ((p - ref_p).abs().max() / ref_p.abs().max()).item() < 5e-5
        for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
            assert abs(report[k] - ref_report[k]) <= 2e-5 * max(abs(ref_report[k]), 1e-3) + 1e-7, (rank, k, report[k], ref_report[k])
    assert all(out[0][2][k] == out[1][2][k] for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"))


def test_direct_rccl_entry_points_one_rank():
    """rlppo_comm_unique_id / rlppo_comm_init / rlppo_allreduce / rlppo_comm_destroy with a one-rank communicator
    (RCCL refuses two ranks on one GPU): fp32 and fp64 buffers come back unchanged, stream-ordered behind the kernel that wrote
    them; errors are reported through the status code."""
    import ctypes
    from rlgym_ppo_amd import _native as N
    L = N.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    buf = torch.zeros(8, device="cuda")
    assert L.rlppo_allreduce(st, ctypes.c_void_p(buf.data_ptr()), 8, 0) != 0          # no communicator yet
    rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(rccl):
        N.check(L.rlppo_comm_set_library(rccl.encode()))
    ident = ctypes.create_string_buffer(N.COMM_ID_BYTES)
    N.check(L.rlppo_comm_unique_id(ident))
    assert L.rlppo_comm_init(1, 1, ident) != 0                                         # rank out of range
    N.check(L.rlppo_comm_init(0, 1, ident))
    assert L.rlppo_comm_init(0, 1, ident) != 0                                         # one communicator per process
    g = torch.randn(341851, device="cuda")
    ref = g.clone()
    g.mul_(2.0)
    N.check(L.rlppo_allreduce(st, ctypes.c_void_p(g.data_ptr()), g.numel(), 0))
    s = torch.arange(8, dtype=torch.float64, device="cuda")
    N.check(L.rlppo_allreduce(st, ctypes.c_void_p(s.data_ptr()), 8, 1))
    torch.cuda.synchronize()
    assert torch.equal(g, ref * 2.0) and torch.equal(s.cpu(), torch.arange(8, dtype=torch.float64))
    N.check(L.rlppo_comm_destroy())
    N.check(L.rlppo_comm_destroy())                                                    # idempotent
