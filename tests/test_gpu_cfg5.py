"""BASELINE configs[4] shape (continuous Gaussian policy, obs 231 = AdvancedObs, 512x4 MLPs, 8 action dims -> 16 outputs),
in fp32 through the generic (non-256-wide) kernel path: K = 256/512, N = 512, 5 linear layers per net."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets, ppo  # noqa: E402
import fp64_gate  # noqa: E402
from test_gpu_kernels import L, relerr, run_minibatch  # noqa: E402,F401


def test_cfg5_shape_minibatch_and_sampling(L):
    torch.manual_seed(5)
    pol = nets.init_mlp(231, (512, 512, 512, 512), 16)
    val = nets.init_mlp(231, (512, 512, 512, 512), 1)
    rs = np.random.RandomState(5)
    n = 6000
    obs = np.clip(rs.randn(n, 231), -5, 5).astype(np.float32)
    mean, std = nets.gauss_out(pol, obs)
    eps = torch.as_tensor(rs.randn(n, 8).astype(np.float32))
    act, logp = nets.gauss_sample(mean, std, eps)
    old = (logp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.1)).numpy()
    adv = rs.randn(n).astype(np.float32)
    tgt = rs.randn(n).astype(np.float32)
    idx = rs.permutation(n)[:3072]
    got = run_minibatch(L, "gaussian", pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, 0.5)
    # float64 truth, every row in; the Gaussian head's (x-mu)^2/sd^3 terms amplify float32 rounding for BOTH float32
    # implementations, which is why the gate is relative to the CPU oracle's own distance from float64
    fp64_gate.gate(L, "gaussian", pol, val, obs[idx], act.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 0.5, got,
                   label="cfg5 shape (obs 231, 512x4, Gaussian head), 3072 rows")


def test_cfg5_policy_classes_round_trip():
    from rlgym_ppo_amd.ppo import ContinuousPolicy, ValueEstimator
    torch.manual_seed(9)
    pol = ContinuousPolicy(231, 16, (512, 512, 512, 512), "cuda:0")
    val = ValueEstimator(231, (512, 512, 512, 512), "cuda:0")
    params = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in pol.arena.linears]
    vparams = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in val.arena.linears]
    rs = np.random.RandomState(1)
    obs = np.clip(rs.randn(4096, 231), -5, 5).astype(np.float32)
    eps = torch.as_tensor(rs.randn(4096, 8).astype(np.float32))
    a, lp = pol.get_action(obs, noise=eps)
    mean, std = nets.gauss_out(params, obs)
    oa, olp = nets.gauss_sample(mean, std, eps)
    np.testing.assert_allclose(a.numpy(), oa.numpy(), rtol=1e-5, atol=3e-6)
    y = pol.arena.forward(pol.arena.stage_obs(obs), out_tanh=True)
    res = fp64_gate.gauss_logp_check(params, obs, eps, y, a, lp, label="cfg5 shape (512x4)")   # float64 truth, derived bound
    assert (lp.numpy() - olp.numpy()).__abs__().max() <= res["hip"][2] * (res["hip"][1] + res["cpu"][1]) + 1e-7
    assert relerr(val(obs), nets.value_forward(vparams, obs)) < 1e-5
    pol.noise_mode = "device"  # fast mode: same distribution, torch's HIP generator
    a2, lp2 = pol.get_action(obs)
    assert a2.shape == (4096, 8) and torch.isfinite(lp2).all() and (a2.abs() <= 1).all()


def test_cfg5_bf16_forward_mode():
    """configs[4] "bf16 fwd / fp32 master weights": the optional rollout precision against its own restatement
    (bf16-rounded operands, fp32 accumulation), and how far it sits from the fp32 forward."""
    from rlgym_ppo_amd.engine import set_inference_precision
    from rlgym_ppo_amd.ppo import ContinuousPolicy, ValueEstimator
    torch.manual_seed(9)
    pol = ContinuousPolicy(231, 16, (512, 512, 512, 512), "cuda:0")
    val = ValueEstimator(231, (512, 512, 512, 512), "cuda:0")
    params = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in pol.arena.linears]
    vparams = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in val.arena.linears]
    rs = np.random.RandomState(2)
    obs = np.clip(rs.randn(4096 + 37, 231), -5, 5).astype(np.float32)  # ragged last row tile
    eps = torch.as_tensor(rs.randn(len(obs), 8).astype(np.float32))
    a32, _ = pol.get_action(obs, noise=eps)
    v32 = val(obs).cpu()
    set_inference_precision("bf16")
    try:
        a16, lp16 = pol.get_action(obs, noise=eps)
        v16 = val(obs).cpu()
    finally:
        set_inference_precision("fp32")
    y = nets.mlp_bf16_operands(params, obs, out_act="tanh")
    m, b = nets.var_map(0.1, 1.0)
    oa, olp = nets.gauss_sample(y[:, :8], y[:, 8:] * m + b, eps)
    # Two-level check: a hidden activation that sits within fp32 summation noise of a bf16 rounding boundary is rounded
    # the other way by one of the two implementations (one bf16 ulp = 0.4 % of that activation); that reaches the
    # outputs at ~1e-4.  So: the bulk agrees to fp32 accuracy, every element to well inside one bf16 ulp of the output.
    da = (a16 - oa).abs()
    assert (da <= 5e-6 + 2e-5 * oa.abs()).float().mean().item() > 0.97 and da.max().item() < 2e-3
    dl = (lp16 - olp).abs()
    assert (dl <= 1e-3 + 1e-4 * olp.abs()).float().mean().item() > 0.97 and dl.max().item() < 0.5
    # value head: hidden layers through the bf16-operand GEMMs, the one-output head stays an fp32 matrix-vector product
    h = torch.as_tensor(obs)
    for w, bb in vparams[:-1]:
        h = torch.relu(torch.nn.functional.linear(h.bfloat16().float(), w.bfloat16().float(), bb))
    ov = torch.nn.functional.linear(h, *vparams[-1])
    dv = (v16 - ov).abs()
    assert (dv <= 1e-6 + 2e-5 * ov.abs()).float().mean().item() > 0.9 and relerr(v16, ov) < 5e-3
    d_a, d_v = (a16 - a32).abs().max().item(), relerr(v16, v32)
    assert 1e-5 < d_a < 0.1 and 1e-5 < d_v < 0.1, (d_a, d_v)  # a different arithmetic, not a different function
    a_back, _ = pol.get_action(obs, noise=eps)
    assert torch.equal(a_back, a32)  # switching back restores the parity mode
