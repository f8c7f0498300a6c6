"""Generality of the kernel paths: odd layer widths (padding to 32 / 64 / 96 / 128 / k*128), one hidden layer, wide
discrete heads (EPL = 8 and 32 register variants of the wave-per-row kernels), tiny and ragged batches, empty calls."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets, ppo  # noqa: E402
import fp64_gate  # noqa: E402
from test_gpu_kernels import L, Net, P, check, dev, relerr, run_minibatch, stream  # noqa: E402,F401


@pytest.mark.parametrize("d,hidden,A,n,mb", [(50, (100, 40), 300, 700, 333), (7, (33,), 5, 64, 64), (300, (130, 257, 64), 1500, 400, 129),
                                             (107, (256, 256, 256), 90, 1300, 1100)])
def test_discrete_odd_shapes(L, d, hidden, A, n, mb):
    torch.manual_seed(d + A)
    pol = nets.init_mlp(d, hidden, A)
    val = nets.init_mlp(d, hidden[::-1], 1)
    rs = np.random.RandomState(n)
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    q = nets.draw_exp_noise(n, A)
    oprobs = nets.discrete_probs(pol, obs)
    oact, ologp = nets.discrete_sample(oprobs, q)
    net = Net(L, pol)
    rows = net.pad(obs)
    act = torch.empty(n, dtype=torch.int64, device="cuda")
    logp = torch.empty(n, device="cuda")
    probs = torch.empty(n, A, device="cuda")
    w = net.ws(n)
    qd = dev(q)
    check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, n, P(qd), P(act), P(logp),
                                  P(probs), P(w), w.numel(), None))
    assert relerr(probs, oprobs) < 1e-5
    assert (act.cpu() != oact).sum().item() <= max(1, n // 500)
    old = (ologp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.2)).numpy()
    adv, tgt = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    idx = rs.permutation(n)[:mb]
    got = run_minibatch(L, "discrete", pol, val, obs, oact.numpy(), old, tgt, adv, idx, 0.2, 0.005, 1.0)
    fp64_gate.gate(L, "discrete", pol, val, obs[idx], oact.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 1.0, got,
                   label=f"odd shape d={d} hidden={hidden} A={A}")


def test_single_row_and_empty_calls(L):
    torch.manual_seed(3)
    pol = nets.init_mlp(107, (64, 64), 90)
    net = Net(L, pol)
    obs = np.random.RandomState(0).randn(1, 107).astype(np.float32)
    rows = net.pad(obs)
    q = nets.draw_exp_noise(1, 90)
    qd = dev(q)
    act = torch.empty(1, dtype=torch.int64, device="cuda")
    logp = torch.empty(1, device="cuda")
    w = net.ws(1)
    check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, 1, P(qd), P(act), P(logp), None,
                                  P(w), w.numel(), None))
    oact, ologp = nets.discrete_sample(nets.discrete_probs(pol, obs), q)
    assert act.item() == oact.item() and abs(logp.item() - ologp.item()) < 1e-5
    # n == 0 is a no-op for every entry point that takes a row count
    check(L, L.rlppo_discrete_act(stream(), net.dims_c, net.nl, P(net.packed), None, net.ld_in, 0, None, None, None, None, None, 0, None))
    check(L, L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), None, net.ld_in, 0, 0, None, net.ld_out, None, 0, None))
    check(L, L.rlppo_pad_rows(stream(), None, 0, 0, 107, 107, None, 128, 0, 0.0, 1.0))
    check(L, L.rlppo_categorical_select(stream(), None, 90, 0, 90, None, None, None))
    # argument errors are reported, not crashed on
    assert L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), P(rows), 64, 1, 0, P(logp), net.ld_out, P(w), w.numel(), None) == 1001
    assert b"ld_obs" in L.rlppo_last_error()
    assert L.rlppo_mlp_forward(stream(), net.dims_c, net.nl, P(net.packed), P(rows), net.ld_in, 1, 0, P(logp), net.ld_out, P(w), 16, None) == 1002


def test_obs_standardisation_fused_in_staging(L):
    # rlppo_pad_rows with the reference's scalar statistics (quirk Q5) == np.clip((obs - mean0) / std0, -5, 5) in fp32
    rs = np.random.RandomState(4)
    obs = (rs.randn(300, 107) * 4 + 1).astype(np.float32)
    mean0, std0 = np.float32(0.73), np.float32(2.9)
    src = dev(obs)
    out = torch.empty(300, 128, device="cuda")
    check(L, L.rlppo_pad_rows(stream(), P(src), 0, 300, 107, 107, P(out), 128, 1, float(mean0), float(std0)))
    ref = np.clip((obs - mean0) / std0, -5, 5)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :107], ref) and (got[:, 107:] == 0).all()


def test_obs_standardisation_per_feature(L):
    # rlppo_pad_rows_per_feature == np.clip((obs - mean) / std, -5, 5) with one (mean, std) per feature, f32 and f64 sources;
    # through NetArena.stage_obs as the managers call it
    rs = np.random.RandomState(5)
    obs = (rs.randn(300, 107) * 4 + 1).astype(np.float32)
    mean = rs.randn(107).astype(np.float32)
    std = (rs.rand(107) * 3 + 0.1).astype(np.float32)
    ref = np.clip((obs - mean) / std, -5, 5)
    mean_d, std_d = dev(mean), dev(std)   # kept alive: the library borrows the pointers
    for is64, src in ((0, dev(obs)), (1, torch.as_tensor(obs.astype(np.float64)).cuda())):
        out = torch.empty(300, 128, device="cuda")
        check(L, L.rlppo_pad_rows_per_feature(stream(), P(src), is64, 300, 107, 107, P(out), 128, P(mean_d), P(std_d)))
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :107], ref) and (got[:, 107:] == 0).all()
    assert L.rlppo_pad_rows_per_feature(stream(), P(src), 0, 300, 107, 107, P(out), 128, None, P(std_d)) != 0
    from rlgym_ppo_amd.ppo import ValueEstimator
    arena = ValueEstimator(107, (32,), "cuda:0").arena
    rows = arena.stage_obs(obs, (torch.from_numpy(mean), torch.from_numpy(std)))
    assert np.array_equal(rows.cpu().numpy()[:, :107], ref)
