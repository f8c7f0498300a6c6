"""world_size-2 gloo test of the data-parallel path (CPU): the product's slice dealing (rlgym_ppo_amd.dp) + one
all-reduce of the flat gradient buffer reproduce the single-rank update.  Gradients come from the CPU oracle here
(there is no GPU in this container); on the GPU box the same dp helpers carry librlppo's gradient arena over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    from oracle import nets
    torch.manual_seed(5)
    pol = nets.init_mlp(20, (16, 16), 6)
    val = nets.init_mlp(20, (16, 16), 1)
    rs = np.random.RandomState(2)
    n = 256
    obs = torch.as_tensor(rs.randn(n, 20).astype(np.float32))
    probs = nets.discrete_probs(pol, obs)
    act, logp = nets.discrete_sample(probs, nets.draw_exp_noise(n, 6))
    buf = dict(states=obs, actions=act.float(), log_probs=logp + 0.1 * torch.as_tensor(rs.randn(n).astype(np.float32)),
               values=torch.as_tensor(rs.randn(n).astype(np.float32)), advantages=torch.as_tensor(rs.randn(n).astype(np.float32)))
    return pol, val, buf


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import nets, ppo
    from rlgym_ppo_amd import dp
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    d, r, w = dp.dist_info()
    assert (r, w) == (rank, world)
    per = 8 // world
    assert dp.slices_for_rank(8, rank, world) == list(range(rank * per, (rank + 1) * per))   # contiguous blocks
    assert dp.slices_for_rank(7, rank, world) == list(range(rank, 7, world))                   # uneven: round-robin
    assert dp.fuse_runs(dp.slices_for_rank(8, rank, world), 8) == [(rank * per, per)]
    assert dp.fuse_runs([0, 1, 2, 3, 4, 5], 4) == [(0, 3), (3, 3)] and dp.fuse_runs([0, 2, 4], 8) == [(0, 1), (2, 1), (4, 1)]
    pol, val, buf = _make_problem()

    class DealtLearn:  # the oracle's loop with the PRODUCT's dealing + collective plugged in
        pass

    report, _, _ = ppo.learn("discrete", pol, val, buf, 128, 32, 2, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(9),
                             rank=rank, world=world, allreduce=lambda t: dp.all_reduce_sum(t, d))
    # ppo.learn deals slices in contiguous blocks when they divide evenly -- assert that this is the product's dealing
    for j in range(4):
        assert (j // (4 // world) == rank) == (j in dp.slices_for_rank(4, rank, world))
    out[rank] = (nets.flatten(pol).clone(), nets.flatten(val).clone(), report)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_equals_single_rank():
    from oracle import nets, ppo
    torch.set_num_threads(1)
    pol, val, buf = _make_problem()
    ref_report, _, _ = ppo.learn("discrete", pol, val, buf, 128, 32, 2, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(9))
    ref_p, ref_v = nets.flatten(pol), nets.flatten(val)
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in (0, 1):
        p, v, report = out[rank]
        # summation order differs (two partial sums added by the collective): rel 1e-5
        assert ((p - ref_p).abs().max() / ref_p.abs().max()).item() < 1e-5
        assert ((v - ref_v).abs().max() / ref_v.abs().max()).item() < 1e-5
        for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
            assert abs(report[k] - ref_report[k]) <= 1e-5 * max(abs(ref_report[k]), 1e-3), k
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])  # replicas stay bit-identical


def test_virtual_ranks_driver_sums_in_rank_order_and_keeps_lock_step():
    """dp.run_virtual_ranks (N ranks in one process: how BASELINE configs[3]'s 8-way partition runs on a one-GPU box) against
    stand-in replicas: every exchange hands each replica the rank-ordered sum, the generators' return values come back per rank,
    and a replica that stops exchanging early is reported instead of deadlocking or being summed short."""
    from rlgym_ppo_amd import dp

    class Replica:
        def __init__(self, rank, steps):
            self.rank, self.steps, self.seen = rank, steps, []

        def learn_steps(self, buf, rank, world):
            assert rank == self.rank and world == 4
            for s in range(self.steps):
                g = torch.full((5,), float(10 ** rank + s))
                yield g
                self.seen.append(g.clone())
            return {"rank": rank, "buf": buf}

    reps = [Replica(r, 3) for r in range(4)]
    out = dp.run_virtual_ranks(reps, ["b%d" % r for r in range(4)])
    assert out == [{"rank": r, "buf": "b%d" % r} for r in range(4)]
    for s in range(3):
        want = torch.full((5,), float(sum(10 ** r + s for r in range(4))))
        assert all(torch.equal(rep.seen[s], want) for rep in reps)
    with pytest.raises(RuntimeError, match="out of step"):
        dp.run_virtual_ranks([Replica(0, 3), Replica(1, 2), Replica(2, 3), Replica(3, 3)], [None] * 4)
