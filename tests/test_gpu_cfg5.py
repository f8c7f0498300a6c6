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


@pytest.mark.parametrize("shape", ["cfg5", "cfg2", "odd"])
def test_bf16_update_precision_against_its_restatement(L, shape):
    """BASELINE configs[4] "bf16 fwd / fp32 master weights" inside the UPDATE (rlppo_set_update_precision(1)): mixed-precision
    training as torch writes it.  Every forward product multiplies bf16-rounded operands (bf16 MFMA, fp32 accumulate); the hidden
    activations are bf16 tensors, so the gradient with respect to each of them is rounded to bf16 too and every backward product
    (dX, dW) multiplies bf16 values on the bf16 MFMA with fp32 accumulation; loss, dW / db accumulation, clip and Adam stay fp32
    on the fp32 master weights.  Checked against oracle/ppo.py::minibatch_autograd under oracle/nets.py::bf16_operands -- torch
    autograd of F.linear(h.bfloat16().float(), r(W), b), r = rounding of a master weight with an identity backward.
    Two correct evaluations of that function agree up to fp32 summation order EXCEPT where an activation (or a gradient) sits within
    that noise of a bf16 rounding boundary and is rounded the other way (one bf16 ulp = 0.4 %), which shifts the next layer's sums
    and cascades.  The test MEASURES that floor -- the restatement against itself with its sums taken in float64
    (oracle/nets.py::sum64): ~4e-4 at 256x3, ~2e-2 at 512x4 with the ill-conditioned Gaussian head -- and holds the kernels
    to 3x it.  Also printed: the distance of the mode from float64 truth of the unrounded function (10-15 % of the gradient: a
    different arithmetic), and the fp32 parity mode is checked to be untouched afterwards.  "odd" = widths the bf16 kernels do
    not cover (fp32 kernels on fp32 copies of the same bf16 values + the same rounding passes: identical mathematics)."""
    torch.manual_seed(11)
    rs = np.random.RandomState(11)
    if shape == "cfg5":
        d, hid, head, n, mb = 231, (512, 512, 512, 512), "gaussian", 5000, 3072
    elif shape == "cfg2":
        d, hid, head, n, mb = 107, (256, 256, 256), "discrete", 5000, 3000
    else:
        d, hid, head, n, mb = 50, (96, 128, 40), "discrete", 900, 700
    n_out = 16 if head == "gaussian" else 90
    pol, val = nets.init_mlp(d, hid, n_out), nets.init_mlp(d, hid, 1)
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    if head == "gaussian":
        mean, std = nets.gauss_out(pol, obs)
        act, logp = nets.gauss_sample(mean, std, torch.as_tensor(rs.randn(n, 8).astype(np.float32)))
    else:
        act, logp = nets.discrete_sample(nets.discrete_probs(pol, obs), nets.draw_exp_noise(n, n_out))
        act = act.float()
    old = (logp + torch.as_tensor(rs.randn(n).astype(np.float32) * 0.1)).numpy()
    adv, tgt = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    idx = rs.permutation(n)[:mb]
    ti = torch.as_tensor(idx)
    gp, gv, st = run_minibatch(L, head, pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, 0.5, precision="bf16")
    with nets.bf16_operands():
        ref = ppo.minibatch_autograd(head, pol, val, torch.as_tensor(obs)[ti], act[ti], torch.as_tensor(old)[ti], torch.as_tensor(adv)[ti],
                                     torch.as_tensor(tgt)[ti], 0.2, 0.005, 0.5)
    with nets.bf16_operands(), nets.sum64():
        ref2 = ppo.minibatch_autograd(head, pol, val, torch.as_tensor(obs)[ti], act[ti], torch.as_tensor(old)[ti], torch.as_tensor(adv)[ti],
                                      torch.as_tensor(tgt)[ti], 0.2, 0.005, 0.5)
    floor = fp64_gate.grads_err(ref2["grad_policy"] + ref2["grad_value"], ref["grad_policy"] + ref["grad_value"])
    truth = ppo.minibatch_analytic(head, pol, val, obs[idx], act.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 0.5)
    e_ref = fp64_gate.grads_err(gp + gv, ref["grad_policy"] + ref["grad_value"])
    e_true = fp64_gate.grads_err(gp + gv, truth["grad_policy"] + truth["grad_value"])
    e_ref_true = fp64_gate.grads_err(ref["grad_policy"] + ref["grad_value"], truth["grad_policy"] + truth["grad_value"])
    print(f"[bf16 update] {shape}: err(HIP bf16, CPU bf16 restatement)={e_ref:.2e}   summation-order floor of the restatement itself="
          f"{floor:.2e}   distance from float64 truth of the fp32 function: HIP {e_true:.2e}, restatement {e_ref_true:.2e}")
    per = [max(fp64_gate._rel(gw, ww), fp64_gate._rel(gb, wb)) for (gw, gb), (ww, wb) in zip(gp + gv, ref["grad_policy"] + ref["grad_value"])]
    print("   per-layer err vs restatement (policy layers, then critic):", " ".join(f"{e:.1e}" for e in per),
          " stats HIP", [float(f"{x:.6g}") for x in st[:5]], "ref", [float(f"{ref[k]:.6g}") for k in ("entropy", "kl", "value_loss", "clip_fraction", "policy_loss")])
    assert e_ref <= max(1e-3, 3.0 * floor), (e_ref, floor)
    assert 1e-4 < e_true < 0.5 and abs(e_true - e_ref_true) < 0.1 * e_ref_true + 3.0 * floor + 2e-3
    for name, k in (("entropy", 0), ("kl", 1), ("value_loss", 2), ("policy_loss", 4)):
        tol = max(2e-4, 3.0 * abs(ref2[name] - ref[name]) / max(abs(ref[name]), 1e-2))
        assert abs(st[k] - ref[name]) <= tol * max(abs(ref[name]), 1e-2), (name, st[k], ref[name], ref2[name])
    # the parity mode is back
    got = run_minibatch(L, head, pol, val, obs, act.numpy(), old, tgt, adv, idx, 0.2, 0.005, 0.5)
    fp64_gate.gate(L, head, pol, val, obs[idx], act.numpy()[idx], old[idx], adv[idx], tgt[idx], 0.2, 0.005, 0.5, got, label=f"fp32 after bf16 ({shape})")


def test_bf16_update_precision_is_additive_at_full_size(L):
    """configs[4] shape at the update's launch size, bf16 update precision: ONE fused pass over 524,288 rows against its eight
    65,536-row minibatches (mb_ratio 1/8 each).  Per-row arithmetic is the same in both (the minibatch scalings are powers of two,
    so every bf16 rounding falls the same way) and a weight gradient is never rounded, so the gradients must agree up to fp32
    summation order -- with the 256 x 256-tile forward / dX / dW kernels, the narrow-head kernels and their reductions at the size
    bench.py --config cfg5 --precision bf16 launches them."""
    torch.manual_seed(41)
    pol = nets.init_mlp(231, (512, 512, 512, 512), 16)
    val = nets.init_mlp(231, (512, 512, 512, 512), 1)
    rs = np.random.RandomState(41)
    n = 524288
    obs = np.clip(rs.randn(n, 231), -5, 5).astype(np.float32)
    act = rs.uniform(-1, 1, size=(n, 8)).astype(np.float32)
    old = (rs.randn(n) * 0.5 - 8.0).astype(np.float32)       # any old log-probabilities: additivity does not care
    adv, tgt = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    idx = rs.permutation(n)
    gp, gv, st = run_minibatch(L, "gaussian", pol, val, obs, act, old, tgt, adv, idx, 0.2, 0.005, 1.0, precision="bf16")
    acc = [[torch.zeros_like(w, dtype=torch.float64), torch.zeros_like(b, dtype=torch.float64)] for w, b in gp + gv]
    st_sum = np.zeros(5)
    for j in range(8):
        gpj, gvj, stj = run_minibatch(L, "gaussian", pol, val, obs, act, old, tgt, adv, idx[j * 65536:(j + 1) * 65536], 0.2, 0.005, 0.125,
                                      precision="bf16")
        for a, g in zip(acc, gpj + gvj):
            a[0] += g[0].double()
            a[1] += g[1].double()
        st_sum += stj[:5]
    assert all(torch.isfinite(g[0]).all() and torch.isfinite(g[1]).all() for g in gp + gv) and float(gp[0][0].abs().max()) > 0
    for whole, parts in zip(gp + gv, acc):
        for k in (0, 1):
            assert relerr(whole[k], parts[k]) < 1e-5
    np.testing.assert_allclose(st[:5], st_sum / 8, rtol=1e-5, atol=1e-8)


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


def test_learner_runs_in_the_bf16_update_precision():
    """PPOLearner.learn under engine.set_update_precision("bf16"): the rounded weight images follow every optimiser step, the
    parameters stay finite and close to the fp32 run's (same data, same shuffle), and switching back restores the fp32 path."""
    from rlgym_ppo_amd.engine import set_update_precision, update_precision
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner

    def run(mode):
        torch.manual_seed(3)
        learner = PPOLearner(231, 8, 2, (512, 512), (512, 512), (0.1, 1.0), 2048, 2, 3e-4, 3e-4, 0.2, 0.005, 1024, "cuda:0")
        rs = np.random.RandomState(3)
        n = 4096
        obs = np.clip(rs.randn(n, 231), -5, 5).astype(np.float32)
        act, logp = learner.policy.get_action(obs, noise=torch.as_tensor(rs.randn(n, 8).astype(np.float32)))
        buf = ExperienceBuffer(n, 3, "cpu")
        z = np.zeros(n, np.float32)
        buf.submit_experience(obs, act.numpy(), logp.numpy() + 0.05 * rs.randn(n).astype(np.float32), z, obs, z, z,
                              rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
        set_update_precision(mode)
        try:
            report = learner.learn(buf)
        finally:
            set_update_precision("fp32")
        return learner.policy.arena.flat.clone(), learner.value_net.arena.flat.clone(), report

    p32, v32, r32 = run("fp32")
    p16, v16, r16 = run("bf16")
    assert update_precision() == "fp32"
    assert torch.isfinite(p16).all() and torch.isfinite(v16).all()
    assert r16["Cumulative Model Updates"] == r32["Cumulative Model Updates"] == 4
    # a different arithmetic (bf16 forward), not a different algorithm: 4 Adam steps move a parameter by at most ~4 lr, so the
    # two runs can be at most ~8 lr apart (a near-zero gradient entry whose sign the rounding flips), and are not identical
    moved = (p32 - p16).abs().max().item()
    assert 0 < moved < 8 * 3e-4 * 1.2
    assert (p32 - p16).abs().mean().item() < 0.25 * 3e-4
    for k in ("Policy Entropy", "Value Function Loss"):
        assert abs(r16[k] - r32[k]) <= 2e-2 * abs(r32[k]), (k, r16[k], r32[k])
    p32b, v32b, _ = run("fp32")
    assert torch.equal(p32, p32b) and torch.equal(v32, v32b)   # the fp32 path is bit-reproducible and untouched by the switch
