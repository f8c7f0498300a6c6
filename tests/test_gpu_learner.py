"""GPU parity of the reference-shaped host classes (PPOLearner / ExperienceBuffer / policies / compute_gae) against
the golden fixtures produced by the reference itself and against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gae as ogae  # noqa: E402
from oracle import nets, ppo  # noqa: E402
import fp64_gate  # noqa: E402  (tests/fp64_gate.py)


def relerr(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_learner(cfg):
    from rlgym_ppo_amd.ppo import PPOLearner
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    return PPOLearner(cfg["d"], cfg["n_act"], cfg["policy_type"], tuple(cfg["layers"]), tuple(cfg["layers"]), (0.1, 1.0),
                      cfg["B"], cfg["epochs"], cfg["lr"], cfg["lr"], cfg["clip"], cfg["ent"], cfg["MB"], "cuda:0")


@pytest.mark.parametrize("name", ["g5_learn_discrete", "g9_learn_continuous", "g9_learn_multidiscrete"])
def test_learn_matches_reference_fixture(golden, name):
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden(name)
    cfg = json.loads(str(g["cfg"]))
    learner = make_learner(cfg)
    # same seed => the reference's initial weights, bit for bit (nn.Linear init consumed in the same order)
    sd = learner.policy.state_dict()
    for k, v in sd.items():
        assert torch.equal(v.cpu(), torch.as_tensor(g["p0." + k])), k
    for k, v in learner.value_net.state_dict().items():
        assert torch.equal(v.cpu(), torch.as_tensor(g["v0." + k])), k

    buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    buf.submit_experience(*[g["exp." + k] for k in names])

    # step-by-step parameter snapshots: run learn() with n_epochs=1 repeatedly (the buffer's generator persists, so the
    # permutation stream is the reference's; Adam state persists in the optimisers)
    n_steps = int(g["n_steps"])
    steps_per_epoch = cfg["n"] // cfg["B"]
    # FLOAT64 TRUTH of the same learn() (oracle/ppo.py::learn64: float64 gradients, clip, Adam) after every optimiser step:
    # the HIP parameters and the reference's own float32 parameters (the fixture) are both measured against it
    head = {0: "discrete", 1: "multidiscrete", 2: "gaussian"}[cfg["policy_type"]]
    truth, weakest = {}, {}
    ppo.learn64(head, nets.params_from_state(g, "p0."), nets.params_from_state(g, "v0."),
                {k: g["exp." + k] for k in ("states", "actions", "log_probs", "values", "advantages")}, cfg["B"], cfg["MB"],
                cfg["epochs"], cfg["clip"], cfg["ent"], cfg["lr"], cfg["lr"], np.random.RandomState(cfg["seed"]), weakest=weakest,
                on_step=lambda i, p, v: truth.__setitem__(i, (np.concatenate([t.ravel() for wb in p for t in wb]),
                                                              np.concatenate([t.ravel() for wb in v for t in wb]),
                                                              weakest["pol"].copy(), weakest["val"].copy())))
    learner.n_epochs = 1
    reports = []
    for e in range(cfg["epochs"]):
        reports.append(learner.learn(buf))
        s = (e + 1) * steps_per_epoch - 1
        pv = torch.nn.utils.parameters_to_vector(learner.policy.parameters())
        vv = torch.nn.utils.parameters_to_vector(learner.value_net.parameters())
        tp, tv, wp, wv = truth[s]
        # Adam's step lr * m / (sqrt(v) + 1e-8) is scale-free: to first order an absolute error d in a gradient entry g_i moves the
        # step by lr * d / |g_i| (at most by the whole step).  With d = 1e-5 max|g| -- the gradient tolerance of the north star,
        # which the minibatch gate holds both implementations to -- a parameter whose gradient fell to w_i max|g| in some step
        # (w_i = `weakest` of oracle/ppo.py::learn64) may therefore be off by
        #     bound_i = (steps so far) * lr * min(1, 1e-5 / w_i)
        # whatever float32 implementation computed it; for w_i >= 1e-4 that is below a tenth of a step and the plain 1e-5
        # parameter gate applies, below (2-3 % of the parameters) the DERIVED bound does -- the same one for the HIP kernels and
        # for the reference's own float32 fixture, both reported as fractions of it.  [r3: replaces a flat 5 % of a step.]
        errs = {}
        for who, p_, v_ in (("hip", pv.detach().cpu().numpy().astype(np.float64), vv.detach().cpu().numpy().astype(np.float64)),
                            ("ref", g[f"step{s}.policy"].astype(np.float64), g[f"step{s}.value"].astype(np.float64))):
            worst_good = worst_ill = frac_ill = 0.0
            for got, tr, weak in ((p_, tp, wp), (v_, tv, wv)):
                d = np.abs(got - tr)
                ill = weak < 1e-4
                worst_good = max(worst_good, float(d[~ill].max() / np.abs(tr).max()))
                if ill.any():
                    bound = (s + 1) * cfg["lr"] * np.minimum(1.0, 1e-5 / np.maximum(weak[ill], 1e-300))
                    worst_ill = max(worst_ill, float(d[ill].max() / np.abs(tr).max()))
                    frac_ill = max(frac_ill, float((d[ill] / bound).max()))
            errs[who] = (worst_good, worst_ill, frac_ill)
        n_ill = int((wp < 1e-4).sum() + (wv < 1e-4).sum())
        print(f"[fp64 gate] {name} after optimiser step {s}: err(HIP, fp64)={errs['hip'][0]:.2e}  err(reference fp32 fixture, fp64)="
              f"{errs['ref'][0]:.2e}  | {n_ill} of {wp.size + wv.size} parameters with an ill-conditioned Adam step: HIP "
              f"{errs['hip'][1]:.1e} = {errs['hip'][2]:.3f} of the derived bound, reference {errs['ref'][1]:.1e} = {errs['ref'][2]:.3f}")
        assert errs["hip"][0] <= max(1e-5, 1.5 * errs["ref"][0]), (name, s, errs)
        assert n_ill <= 0.05 * (wp.size + wv.size)
        assert errs["hip"][2] <= 1.0 and errs["ref"][2] <= 1.0, (name, s, errs)
    assert learner.cumulative_model_updates == n_steps
    assert sorted(reports[0].keys()) == sorted([
        "PPO Batch Consumption Time", "Cumulative Model Updates", "Policy Entropy", "Mean KL Divergence",
        "Value Function Loss", "SB3 Clip Fraction", "Policy Update Magnitude", "Value Function Update Magnitude"])
    # the fixture's report averages over both epochs
    for key in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
        got = float(np.mean([r[key] for r in reports]))
        ref = float(g["report." + key])
        assert abs(got - ref) <= 2e-5 * max(abs(ref), 1e-4) + 1e-7, (key, got, ref)
    # optimiser state in the stock layout
    osd = learner.policy_optimizer.state_dict()
    assert float(osd["state"][0]["step"]) == float(g["adam_step"])
    assert relerr(osd["state"][0]["exp_avg"], g["adam_exp_avg0"]) < 1e-4
    assert relerr(osd["state"][0]["exp_avg_sq"], g["adam_exp_avg_sq0"]) < 1e-4


@pytest.mark.parametrize("precision", ["fp32", "x3"])
def test_learn_at_the_paired_launch_size_matches_the_reference_fixture(golden, precision):
    """(precision "x3": the same fixture and the same gates with rlppo_set_update_precision(2) -- the hidden forward / dX products
    on the bf16 MFMA pipe from three-piece operands, policy and critic as two chains)
    [r4] G5big: the reference's PPOLearner.learn at 256x3, n = B = 262,144, MB = 65,536, 2 epochs -- the size from which the update
    runs its PAIRED policy + critic launches with the minibatch gather FUSED into the first layer (csrc/api.hip), which until
    round 4 no reference-held vector ever reached.  The experience is rebuilt from seeds by the generator's own function (its bit
    hashes are in the fixture), the seeded construction must give the reference's initial parameters bit for bit (hash + first 64
    values), a library counter says which launch form ran, and the parameters after both optimiser steps go through the same
    float64-yardstick gate as the small learn() fixtures: HIP and the reference's own float32 result, both against
    oracle/ppo.py::learn64.  The sharp comparison is the first step's batch GRADIENT (what the north star's 1e-5 is about; the
    fixture holds every 8th entry + norms of what the reference handed to clip_grad_norm_): err(HIP, fp64) <= max(1e-5, 1.5 x
    err(reference, fp64)) of max|g|.  Parameters after Adam show a gradient only through a scale-free step: on this workload
    (uniform actions, noise advantages: the gradient is a few outliers -- the head's biases -- over a sea of cancelling sums, 13 %
    of the entries below 1e-4 of the largest) the reference's own float32 parameters sit 1.5e-3 of max|p| from float64 truth
    after two steps, so the parameter gate is the yardstick alone: well-conditioned and ill-conditioned parameters each within
    1.5 x the reference's own distance (the derived-bound fractions are printed for both)."""
    import importlib.util
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    L = N.lib()
    g = golden("g5big_learn_discrete_256x3")
    cfg = json.loads(str(g["cfg"]))
    learner = make_learner(cfg)
    vec = lambda m: torch.nn.utils.parameters_to_vector(m.parameters()).detach().cpu().numpy()
    p0, v0 = vec(learner.policy), vec(learner.value_net)
    assert mg.param_hash(p0) == g["p0.hash"] and mg.param_hash(v0) == g["v0.hash"]      # the reference's initial weights, bit for bit
    assert np.array_equal(p0[:64], g["p0.head"]) and np.array_equal(v0[:64], g["v0.head"])
    # (the experience and its float64 truth are computed once per session: the fp32 and x3 parametrisations share them)
    if "exp" not in _G5BIG:
        _G5BIG["exp"] = mg.g5big_inputs(cfg)
    exp = _G5BIG["exp"]
    assert np.array_equal(np.asarray([mg.param_hash(x.reshape(-1)) for x in exp[:3] + exp[7:]]), g["exp.hash"])   # the same experience
    buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    buf.submit_experience(*exp)
    _G5BIG["v0"] = v0

    # float64 truth of both optimiser steps (CPU oracle, ~1 TFLOP of float64)
    layers = [cfg["d"]] + list(cfg["layers"])
    def split(flat, outs):
        params, o = [], 0
        dims = layers + [outs]
        for i in range(len(dims) - 1):
            w = flat[o:o + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]); o += w.size
            b = flat[o:o + dims[i + 1]]; o += b.size
            params.append((torch.as_tensor(w.copy()), torch.as_tensor(b.copy())))
        return params
    if "truth" not in _G5BIG:
        truth, weakest, grad64 = {}, {}, {}
        ppo.learn64("discrete", split(p0, cfg["n_act"]), split(v0, 1),
                    dict(states=exp[0], actions=exp[1], log_probs=exp[2], values=exp[7], advantages=exp[8]), cfg["B"], cfg["MB"], cfg["epochs"],
                    cfg["clip"], cfg["ent"], cfg["lr"], cfg["lr"], np.random.RandomState(cfg["seed"]), weakest=weakest,
                    on_grad=lambda i, gp, gv: grad64.__setitem__(i, (gp.copy(), gv.copy())),
                    on_step=lambda i, p, v: truth.__setitem__(i, (np.concatenate([t.ravel() for wb in p for t in wb]),
                                                                  np.concatenate([t.ravel() for wb in v for t in wb]),
                                                                  weakest["pol"].copy(), weakest["val"].copy())))
        _G5BIG["truth"], _G5BIG["grad64"] = truth, grad64
    truth, grad64 = _G5BIG["truth"], _G5BIG["grad64"]
    n_steps = int(g["n_steps"])
    assert n_steps == 2 and sorted(truth) == [0, 1]
    learner.n_epochs = 1
    from rlgym_ppo_amd.engine import set_update_precision
    set_update_precision(precision)
    try:
        _g5big_body(L, g, cfg, learner, buf, truth, grad64, n_steps, p0, vec, precision)
    finally:
        set_update_precision("fp32")


_G5BIG = {}  # the G5big experience / critic parameters of the running test and, once measured, the worth D of its batch's ReLU decisions


def _g5big_body(L, g, cfg, learner, buf, truth, grad64, n_steps, p0, vec, precision):
    hip_grads = []
    learner.grad_probe = lambda gr: hip_grads.append(gr.detach().cpu().numpy().astype(np.float64))
    passes0, paired0, gfused0 = (int(L.rlppo_dbg_counter(k)) for k in (2, 3, 4))
    reports = []
    for s in range(n_steps):
        reports.append(learner.learn(buf))
        assert learner._fused_rows == 262144                               # the 4 minibatches of a batch in ONE pass
        pv, vv = vec(learner.policy).astype(np.float64), vec(learner.value_net).astype(np.float64)
        tp, tv, wp, wv = truth[s]
        bound = lambda weak: np.where(weak < 1e-4, (s + 1) * cfg["lr"] * np.minimum(1.0, 1e-5 / np.maximum(weak, 1e-300)), 0.0)
        if s < n_steps - 1:   # heads + sums only
            for got, ref_head, ref_sum, tr, weak in ((pv, g[f"step{s}.policy_head"], float(g[f"step{s}.policy_sum"]), tp, wp),
                                                     (vv, g[f"step{s}.value_head"], float(g[f"step{s}.value_sum"]), tv, wv)):
                scale = np.abs(tr).max()
                tol = np.maximum(1e-5 * scale, bound(weak))
                e_hip, e_ref = np.abs(got[:64] - tr[:64]), np.abs(ref_head.astype(np.float64) - tr[:64])
                assert (e_hip <= np.maximum(tol[:64], 1.5 * e_ref.max())).all(), (s, e_hip.max(), e_ref.max())
                # a plain sum moves by at most the sum of the per-parameter allowances
                d_hip, d_ref = abs(got.sum() - tr.sum()), abs(ref_sum - tr.sum())
                print(f"[fp64 gate] g5big ({precision}) step {s}: head err HIP {e_hip.max() / scale:.2e} reference {e_ref.max() / scale:.2e} of max|p|; "
                      f"|sum - sum64| HIP {d_hip:.3e} reference {d_ref:.3e} (allowance {tol.sum():.3e})")
                assert d_hip <= max(tol.sum(), 1.5 * d_ref)
            continue
        errs = {}
        for who, p_, v_ in (("hip", pv, vv), ("ref", g[f"step{s}.policy"].astype(np.float64), g[f"step{s}.value"].astype(np.float64))):
            worst_good = worst_ill = frac_ill = 0.0
            for got, tr, weak in ((p_, tp, wp), (v_, tv, wv)):
                d = np.abs(got - tr)
                ill = weak < 1e-4
                worst_good = max(worst_good, float(d[~ill].max() / np.abs(tr).max()))
                if ill.any():
                    worst_ill = max(worst_ill, float(d[ill].max() / np.abs(tr).max()))
                    frac_ill = max(frac_ill, float((d[ill] / bound(weak)[ill]).max()))
            errs[who] = (worst_good, worst_ill, frac_ill)
        n_ill = int((wp < 1e-4).sum() + (wv < 1e-4).sum())
        print(f"[fp64 gate] g5big_learn_discrete_256x3 ({precision}) after optimiser step {s}: err(HIP, fp64)={errs['hip'][0]:.2e}  err(reference fp32 "
              f"fixture, fp64)={errs['ref'][0]:.2e}  | {n_ill} of {wp.size + wv.size} parameters with an ill-conditioned Adam step: HIP "
              f"{errs['hip'][1]:.1e} = {errs['hip'][2]:.3f} of the ARITHMETIC bound (d = 1e-5), reference {errs['ref'][1]:.1e} = {errs['ref'][2]:.3f}")
        assert errs["hip"][0] <= max(1e-5, 1.5 * errs["ref"][0]), errs
        assert errs["hip"][1] <= max(1e-5, 1.5 * errs["ref"][1]), errs
        # [r5] The bound above prices a gradient tolerance of d = 1e-5 -- arithmetic -- and BOTH float32 results exceed it (the
        # reference's own by more than the product's): what separates any float32 evaluation from float64's here is not
        # arithmetic but the ReLU decisions a last bit decides.  With d = D, the measured worth of those decisions on this batch
        # (fp64_gate.gate: the same float64 gradient under the implementation's own masks against float64's own; for the
        # reference, whose masks on the generating host are not observable here, the CPU oracle's on this host stand in), every
        # entry of the product's result is inside (steps) lr min(1, D / w_i) + 1e-5 max|p| -- a derived bound that holds, no
        # yardstick: HIP 0.64-0.66 of it; the reference's own float32 result sits at 1.09 (printed, not asserted: its decisions
        # were taken on another host).
        if "D" not in _G5BIG:
            import fp64_gate as _gate
            n_pol_, lay = p0.size, [cfg["d"]] + list(cfg["layers"])

            def unflat(flat, outs):
                params, o, dims = [], 0, lay + [outs]
                for i in range(len(dims) - 1):
                    w = flat[o:o + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]); o += w.size
                    bb = flat[o:o + dims[i + 1]]; o += bb.size
                    params.append((w, bb))
                return params
            exp_, v0_ = _G5BIG["exp"], _G5BIG["v0"]
            idx = np.random.RandomState(cfg["seed"]).permutation(cfg["n"])[:cfg["B"]]
            tt = lambda ps: [(torch.as_tensor(w.copy()), torch.as_tensor(b.copy())) for w, b in ps]
            res = _gate.gate(L, "discrete", tt(unflat(p0, cfg["n_act"])), tt(unflat(v0_, 1)), exp_[0][idx], exp_[1][idx], exp_[2][idx], exp_[8][idx],
                             exp_[7][idx], cfg["clip"], cfg["ent"], 1.0, (unflat(hip_grads[0][:n_pol_], cfg["n_act"]), unflat(hip_grads[0][n_pol_:], 1), None),
                             label=f"g5big first-step batch gradient ({precision})", x3=precision == "x3")
            _G5BIG["D"] = res["hip"]["ambiguity"] + res["cpu"]["ambiguity"] + 2e-5
        D = _G5BIG["D"]
        for who, p_, v_ in (("HIP", pv, vv), ("reference", g[f"step{s}.policy"].astype(np.float64), g[f"step{s}.value"].astype(np.float64))):
            worst = 0.0
            for got, tr, weak in ((p_, tp, wp), (v_, tv, wv)):
                allow = (s + 1) * cfg["lr"] * np.minimum(1.0, D / np.maximum(weak, 1e-300)) + 1e-5 * np.abs(tr).max()
                worst = max(worst, float((np.abs(got - tr) / allow).max()))
            print(f"[fp64 gate] g5big ({precision}) parameters after step {s}, {who} against float64: worst entry at {worst:.3f} of the decision-priced bound (D = {D:.1e})")
            assert who != "HIP" or worst <= 1.0, (who, worst)
    passes, paired, gfused = (int(L.rlppo_dbg_counter(k)) for k in (2, 3, 4))
    assert passes - passes0 == n_steps and paired - paired0 == (n_steps if precision == "fp32" else 0) and gfused - gfused0 == n_steps, \
        "the paired / gather-fused launches did not run: (passes, paired, gather-fused) = %s" % ((passes - passes0, paired - paired0, gfused - gfused0),)
    # the first step's batch gradient: [grad_policy | grad_value] of the product against the reference's, both against float64
    n_pol = p0.size
    for tag, mine, t64 in (("policy", hip_grads[0][:n_pol], grad64[0][0]), ("value", hip_grads[0][n_pol:], grad64[0][1])):
        ref8 = g[f"grad0.{tag}_every8"].astype(np.float64)
        scale = np.abs(t64).max()
        e_hip_all, e_hip8, e_ref8 = np.abs(mine - t64).max() / scale, np.abs(mine[::8] - t64[::8]).max() / scale, np.abs(ref8 - t64[::8]).max() / scale
        l2_hip, l2_ref, l2_64 = np.sqrt((mine ** 2).sum()), float(g[f"grad0.{tag}_l2"]), np.sqrt((t64 ** 2).sum())
        print(f"[fp64 gate] g5big ({precision}) first-step {tag} gradient: err(HIP, fp64) = {e_hip_all:.2e} of max|g| over all entries, {e_hip8:.2e} over every "
              f"8th; err(reference, fp64) = {e_ref8:.2e} over every 8th; |g|2 HIP {l2_hip:.8e} reference {l2_ref:.8e} float64 {l2_64:.8e}")
        assert e_hip_all <= max(1e-5, 1.5 * e_ref8), (tag, e_hip_all, e_ref8)
        assert abs(l2_hip - l2_64) <= max(1e-5 * l2_64, 1.5 * abs(l2_ref - l2_64))
        assert abs(float(g[f"grad0.{tag}_max"]) - scale) <= 1e-4 * scale
    for key in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
        got, ref = float(np.mean([r[key] for r in reports])), float(g["report." + key])
        # (the clip fraction is a count: a row whose ratio sits within float32 rounding of a clip edge may fall either way -- 4 rows)
        tol = 4.0 / cfg["B"] if key == "SB3 Clip Fraction" else 2e-5 * max(abs(ref), 1e-4) + 1e-7
        assert abs(got - ref) <= tol, (key, got, ref)


def test_g5bigb_paired_launch_gradient_against_the_reference_directly(golden):
    """[r5] G5big-b: the reference's PPOLearner.learn at the paired-launch size (256x3, n = B = 262,144, MB = 65,536, 2 optimiser steps)
    on a WELL-CONDITIONED workload -- actions sampled by the reference's policy at its initial weights, old log-probabilities that
    sample's + N(0, 0.1) kept 5e-4 away from both clip edges at the first step, advantages and targets with a per-action / per-state
    signal (tests/golden/make_golden.py::g5bigb_inputs) -- so that no float64 yardstick is needed: the first step's batch gradient
    (grouped weight-gradient launch, paired forward / dX launches, fused gather: counters checked) is held to the REFERENCE's own
    gradient, 1e-5 of max|g| (north_star's number; the fixture holds every 8th entry + norms), and the parameters after both steps
    to the reference's own parameters, 1e-5 of max|p| on every entry whose Adam steps were well-conditioned in float64 (smallest
    |g_i| / max|g| over the steps >= 1e-4; oracle/ppo.py::learn64 says which -- the others stay within the derived bound)."""
    import importlib.util
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    L = N.lib()
    g = golden("g5bigb_learn_discrete_256x3")
    cfg = json.loads(str(g["cfg"]))
    learner = make_learner(cfg)
    vec = lambda m: torch.nn.utils.parameters_to_vector(m.parameters()).detach().cpu().numpy()
    p0, v0 = vec(learner.policy), vec(learner.value_net)
    assert mg.param_hash(p0) == g["p0.hash"] and mg.param_hash(v0) == g["v0.hash"]      # the reference's initial weights, bit for bit
    exp = mg.g5bigb_inputs(cfg, g["exp.actions_u8"], g["exp.log_probs"])
    assert np.array_equal(np.asarray([mg.param_hash(x.reshape(-1)) for x in exp[:3] + exp[7:]]), g["exp.hash"])   # the same experience
    buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    buf.submit_experience(*exp)
    layers = [cfg["d"]] + list(cfg["layers"])

    def split(flat, outs):
        params, o = [], 0
        dims = layers + [outs]
        for i in range(len(dims) - 1):
            w = flat[o:o + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]); o += w.size
            b = flat[o:o + dims[i + 1]]; o += b.size
            params.append((torch.as_tensor(w.copy()), torch.as_tensor(b.copy())))
        return params
    truth, weakest, grad64 = {}, {}, {}
    ppo.learn64("discrete", split(p0, cfg["n_act"]), split(v0, 1),
                dict(states=exp[0], actions=exp[1], log_probs=exp[2], values=exp[7], advantages=exp[8]), cfg["B"], cfg["MB"], cfg["epochs"],
                cfg["clip"], cfg["ent"], cfg["lr"], cfg["lr"], np.random.RandomState(cfg["seed"]), weakest=weakest,
                on_grad=lambda i, gp, gv: grad64.__setitem__(i, (gp.copy(), gv.copy())),
                on_step=lambda i, p, v: truth.__setitem__(i, (np.concatenate([t.ravel() for wb in p for t in wb]),
                                                              np.concatenate([t.ravel() for wb in v for t in wb]),
                                                              weakest["pol"].copy(), weakest["val"].copy())))
    n_steps = int(g["n_steps"])
    assert n_steps == 2
    hip_grads = []
    learner.grad_probe = lambda gr: hip_grads.append(gr.detach().cpu().numpy().astype(np.float64))
    learner.n_epochs = 1
    c0 = [int(L.rlppo_dbg_counter(k)) for k in (2, 3, 4, 5)]
    reports = [learner.learn(buf) for _ in range(n_steps)]
    assert [int(L.rlppo_dbg_counter(k)) - c for k, c in zip((2, 3, 4, 5), c0)] == [n_steps] * 4, "paired / gather-fused / grouped launches did not run"
    # ---- the first step's batch gradient.  (a) Against float64 UNDER THE PRODUCT'S OWN ReLU DECISIONS (tests/fp64_gate.py, the G4
    # method: a pre-activation within float32 rounding of 0 is decided either way by a correct float32 forward, and one flipped unit
    # moves a row's worth of gradient -- ~1 / B of a unit's, 2e-5 of max|g| here -- for ANY pair of float32 implementations; with the
    # decisions imposed what is left is arithmetic): the batch is the first 262,144 indices of the epoch's permutation, its gradient
    # the mean over all its rows (4 minibatches x MB / B).  (b) Against the REFERENCE's own gradient directly: every 8th entry of
    # the fixture; the two differ by arithmetic + their few differing decisions.
    n_pol = p0.size

    def unflat(flat, outs):
        params, o = [], 0
        dims = layers + [outs]
        for i in range(len(dims) - 1):
            w = flat[o:o + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]); o += w.size
            bb = flat[o:o + dims[i + 1]]; o += bb.size
            params.append((w, bb))
        return params
    idx = np.random.RandomState(cfg["seed"]).permutation(cfg["n"])[:cfg["B"]]
    res = fp64_gate.gate(L, "discrete", split(p0, cfg["n_act"]), split(v0, 1), exp[0][idx], exp[1][idx], exp[2][idx], exp[8][idx], exp[7][idx],
                         cfg["clip"], cfg["ent"], 1.0, (unflat(hip_grads[0][:n_pol], cfg["n_act"]), unflat(hip_grads[0][n_pol:], 1), None),
                         label="g5bigb first-step batch gradient (262,144 rows, paired + gather-fused + grouped launches)")
    print(f"[direct gate] g5bigb: err(HIP, float64 under HIP's decisions) = {res['hip']['err']:.2e}; the CPU float32 oracle's = {res['cpu']['err']:.2e} "
          f"(max over tensors of max|err| / max|g| of the tensor)")
    assert len(res["edge"]) == 0                                    # the workload keeps every ratio away from the clip edges
    for tag, mine, t64 in (("policy", hip_grads[0][:n_pol], grad64[0][0]), ("value", hip_grads[0][n_pol:], grad64[0][1])):
        ref8 = g[f"grad0.{tag}_every8"].astype(np.float64)
        scale = float(g[f"grad0.{tag}_max"])
        dd = np.abs(mine[::8] - ref8) / scale
        e_hip, e_ref = np.abs(mine - t64).max() / scale, np.abs(ref8 - t64[::8]).max() / scale
        l2_hip, l2_ref = np.sqrt((mine ** 2).sum()), float(g[f"grad0.{tag}_l2"])
        print(f"[direct gate] g5bigb first-step {tag} gradient: |HIP - reference| max {dd.max():.2e}, 99.9 % quantile {np.quantile(dd, 0.999):.2e}, rms "
              f"{np.sqrt((dd ** 2).mean()):.2e} of max|g| (every 8th entry); against float64 with ITS OWN decisions: HIP {e_hip:.2e}, reference {e_ref:.2e}; "
              f"|g|2 HIP {l2_hip:.8e} reference {l2_ref:.8e}")
        # two float32 implementations: 1e-5 of max|g| on all but the entries their differing ReLU decisions touch, those within 3e-5
        assert np.quantile(dd, 0.999) <= 1e-5 and np.sqrt((dd ** 2).mean()) <= 2e-6 and dd.max() <= 3e-5, (tag, float(dd.max()))
        assert abs(l2_hip - l2_ref) <= 1e-5 * l2_ref
        assert abs(np.abs(t64).max() - scale) <= 1e-4 * scale
    # ---- parameters after both steps against the reference's own, directly.  Adam's step lr m / (sqrt(v) + eps) is scale-free: a
    # gradient difference d (relative to max|g|) moves the step of an entry whose gradient fell to w_i max|g| in some step by up to
    # lr min(1, d / w_i).  Between two float32 implementations d is NOT their arithmetic (3e-7 / 8e-6 above) but the ReLU decisions
    # they legitimately take differently: D = the measured worth of both sides' decisions (3e-4 of a tensor's gradient at this
    # size) + 2e-5.  Every entry is held to (steps) lr min(1, D / w_i) + 1e-5 max|p| -- no yardstick, no entry excluded.
    s = n_steps - 1
    pv, vv = vec(learner.policy).astype(np.float64), vec(learner.value_net).astype(np.float64)
    tp, tv, wp, wv = truth[s]
    D = res["hip"]["ambiguity"] + res["cpu"]["ambiguity"] + 2e-5
    for tag, got, ref, tr, weak in (("policy", pv, g[f"step{s}.policy"].astype(np.float64), tp, wp), ("value", vv, g[f"step{s}.value"].astype(np.float64), tv, wv)):
        scale = np.abs(ref).max()
        d = np.abs(got - ref)
        bound = (s + 1) * cfg["lr"] * np.minimum(1.0, D / np.maximum(weak, 1e-300)) + 1e-5 * scale
        frac = d / bound
        print(f"[direct gate] g5bigb {tag} parameters after step {s}: |HIP - reference| max {d.max() / scale:.2e} of max|p|, 99 % quantile "
              f"{np.quantile(d, 0.99) / scale:.2e}; worst entry at {frac.max():.3f} of its derived allowance (D = {D:.1e}); against float64 with its own "
              f"decisions: HIP max {np.abs(got - tr).max() / scale:.2e}, reference {np.abs(ref - tr).max() / scale:.2e}")
        assert frac.max() <= 1.0, (tag, float(frac.max()))
        assert np.quantile(d, 0.99) / scale <= 5e-5, (tag, float(np.quantile(d, 0.99) / scale))
    for key in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
        got, ref = float(np.mean([r[key] for r in reports])), float(g["report." + key])
        tol = 4.0 / cfg["B"] if key == "SB3 Clip Fraction" else 2e-5 * max(abs(ref), 1e-4) + 1e-7
        assert abs(got - ref) <= tol, (key, got, ref)


def test_reference_default_batch_of_50000_rows_against_the_oracle():
    """[r5] The reference's own configuration (learner.py:34-53: ppo_batch_size 50,000, minibatch = batch, buffer 100,000; 256x3): one
    learn() = 2 optimiser steps of ONE 50,000-row pass each -- 390.6 row tiles (ragged forward / dX launches), no minibatch fusion,
    no paired launches, the grouped weight-gradient launch on a row count that is no multiple of anything -- against the CPU oracle
    (the reference's op sequence) on identical inputs: parameters, report and the first step's batch gradient (float64 yardstick
    for the gradient; the workload is the well-conditioned one of G5big-b, actions uniform)."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    L = N.lib()
    n, B, d, A, seed = 100_000, 50_000, 107, 90, 77
    torch.manual_seed(seed)
    learner = PPOLearner(d, A, 0, (256, 256, 256), (256, 256, 256), (0.1, 1.0), B, 1, 3e-4, 3e-4, 0.2, 0.005, B, "cuda:0")
    pol0 = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in learner.policy.arena.linears]
    val0 = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in learner.value_net.arena.linears]
    rs = np.random.RandomState(seed)
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    acts = rs.randint(0, A, n)
    with torch.no_grad():
        lp = torch.log(nets.discrete_probs(pol0, torch.as_tensor(obs)))[torch.arange(n), torch.as_tensor(acts)].numpy()
    old = (lp + 0.1 * rs.randn(n)).astype(np.float32)
    adv = (0.5 * rs.randn(A)[acts] + 0.7 * np.sign(obs[np.arange(n), acts]) + 0.3 * rs.randn(n)).astype(np.float32)
    tgt = (1.5 * np.sign(obs[:, 0]) + 0.5 * obs[:, 1] + 0.3 * rs.randn(n)).astype(np.float32)
    z = np.zeros(n, np.float32)
    buf = ExperienceBuffer(n, seed, "cpu")
    buf.submit_experience(obs, acts.astype(np.float32), old, z, obs[:1].repeat(n, 0), z, z, tgt, adv)
    hip_grads = []
    learner.grad_probe = lambda gr: hip_grads.append(gr.detach().cpu().numpy().astype(np.float64))
    c0 = [int(L.rlppo_dbg_counter(k)) for k in (2, 3, 5)]
    report = learner.learn(buf)
    assert learner._fused_rows == B
    assert [int(L.rlppo_dbg_counter(k)) - c for k, c in zip((2, 3, 5), c0)] == [2, 0, 2]   # 2 passes, unpaired, grouped weight gradients
    obuf = dict(states=torch.as_tensor(obs), actions=torch.as_tensor(acts.astype(np.float32)), log_probs=torch.as_tensor(old),
                values=torch.as_tensor(tgt), advantages=torch.as_tensor(adv))
    opol, oval = [(w.clone(), b.clone()) for w, b in pol0], [(w.clone(), b.clone()) for w, b in val0]
    oreport, _, _ = ppo.learn("discrete", opol, oval, obuf, B, B, 1, 0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(seed),
                              on_step=None)
    got_p = torch.nn.utils.parameters_to_vector(learner.policy.parameters()).cpu()
    got_v = torch.nn.utils.parameters_to_vector(learner.value_net.parameters()).cpu()
    ep, ev = relerr(got_p, nets.flatten(opol)), relerr(got_v, nets.flatten(oval))
    print(f"[50,000-row learn()] parameters after 2 optimiser steps, HIP against the CPU oracle: policy {ep:.2e}, critic {ev:.2e} of max|p|")
    # the first step's gradient (one 50,000-row pass: the first 50,000 indices of the epoch's permutation) against float64 under the
    # product's own ReLU decisions (tests/fp64_gate.py: what a float32 rounding may legitimately decide either way is imposed,
    # what is left is arithmetic) -- the same gate, with the same floor, as the minibatch tests
    n_pol = got_p.numel()
    dims_p, dims_v = [d, 256, 256, 256, A], [d, 256, 256, 256, 1]

    def unflat(flat, dims):
        params, o = [], 0
        for i in range(len(dims) - 1):
            w = flat[o:o + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]); o += w.size
            bb = flat[o:o + dims[i + 1]]; o += bb.size
            params.append((w, bb))
        return params
    idx = np.random.RandomState(seed).permutation(n)[:B]
    res = fp64_gate.gate(L, "discrete", pol0, val0, obs[idx], acts[idx].astype(np.float32), old[idx], adv[idx], tgt[idx], 0.2, 0.005, 1.0,
                         (unflat(hip_grads[0][:n_pol], dims_p), unflat(hip_grads[0][n_pol:], dims_v), None), label="50,000-row pass of learn()")
    print(f"[50,000-row learn()] first-step gradient: err(HIP, float64 under HIP's decisions) = {res['hip']['err']:.2e}, CPU float32 oracle {res['cpu']['err']:.2e}")
    # Parameters after both steps against the CPU oracle's, ENTRY BY ENTRY and with no entry excluded (G5big-b's method; round 5 gated
    # quantiles here, which left 2 % of the entries ungated).  Adam's step lr m / (sqrt(v) + eps) is scale-free: a gradient difference
    # d (relative to max|g|) moves the step of an entry whose gradient fell to w_i max|g| in some step by up to lr min(1, d / w_i).
    # Between two float32 implementations d is not their arithmetic (5e-7 / 5e-6 above) but the ReLU decisions and the clip-edge row
    # they legitimately take differently: D = the measured worth of both sides' decisions on this batch + 2e-5; w_i = the entry's
    # smallest |g_i| / max|g| over the two steps in float64 (oracle/ppo.py::learn64).  Every entry is held to
    # (steps) lr min(1, D / w_i) + 1e-5 max|p|.
    truth, weakest = {}, {}
    ppo.learn64("discrete", pol0, val0, dict(states=obs, actions=acts.astype(np.float32), log_probs=old, values=tgt, advantages=adv), B, B, 1,
                0.2, 0.005, 3e-4, 3e-4, np.random.RandomState(seed), weakest=weakest,
                on_step=lambda i, p, v: truth.__setitem__(i, (np.concatenate([t.ravel() for wb in p for t in wb]),
                                                              np.concatenate([t.ravel() for wb in v for t in wb]))))
    assert sorted(truth) == [0, 1]
    D = res["hip"]["ambiguity"] + res["cpu"]["ambiguity"] + 2e-5
    for tag, got, ref, weak, t64 in (("policy", got_p, nets.flatten(opol).detach(), weakest["pol"], truth[1][0]),
                                     ("value", got_v, nets.flatten(oval).detach(), weakest["val"], truth[1][1])):
        got64, ref64 = got.detach().numpy().astype(np.float64), ref.numpy().astype(np.float64)
        scale = float(np.abs(ref64).max())
        dd = np.abs(got64 - ref64)
        allow = 2 * 3e-4 * np.minimum(1.0, D / np.maximum(weak, 1e-300)) + 1e-5 * scale
        frac = dd / allow
        n_ill = int((weak < 1e-4).sum())
        print(f"[50,000-row learn()] {tag} parameters, HIP against the CPU oracle: median {np.median(dd) / scale:.1e}, 98 % {np.quantile(dd, 0.98) / scale:.1e}, "
              f"max {dd.max() / scale:.1e} of max|p|; worst entry at {frac.max():.3f} of its derived allowance (D = {D:.1e}; {n_ill} of {weak.size} entries "
              f"with an ill-conditioned Adam step); against float64: HIP {np.abs(got64 - t64).max() / scale:.1e}, CPU oracle {np.abs(ref64 - t64).max() / scale:.1e}")
        assert frac.max() <= 1.0, (tag, float(frac.max()))
        assert np.median(dd) / scale <= 1e-5, (tag, float(np.median(dd) / scale))
    for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
        tol = 4.0 / B if k == "SB3 Clip Fraction" else 2e-5 * max(abs(oreport[k]), 1e-4) + 1e-7
        assert abs(report[k] - oreport[k]) <= tol, (k, report[k], oreport[k])


def test_two_learners_of_one_process_train_in_different_precisions():
    """[r5] The update precision is an argument of the call (rlppo_minibatch_args.precision <- PPOLearner.update_precision), not only
    a process-wide switch: an fp32 learner and a split-bf16 ("x3") learner of ONE process, their learn() calls interleaved, each end
    bit-identical to the same learner trained alone -- and the process default (fp32) is never touched."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    n, B, MB, d, A = 8192, 4096, 2048, 107, 90
    rs = np.random.RandomState(11)
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    acts = rs.randint(0, A, n).astype(np.float32)
    old = (-np.log(A) + 0.1 * rs.randn(n)).astype(np.float32)
    adv, tgt = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    z = np.zeros(n, np.float32)

    def make(prec):
        torch.manual_seed(3)
        lr_ = PPOLearner(d, A, 0, (256, 256), (256, 256), (0.1, 1.0), B, 1, 3e-4, 3e-4, 0.2, 0.005, MB, "cuda:0")
        lr_.update_precision = prec
        buf = ExperienceBuffer(n, 5, "cpu")
        buf.submit_experience(obs, acts, old, z, obs, z, z, tgt, adv)
        return lr_, buf

    vec = lambda l: torch.cat((torch.nn.utils.parameters_to_vector(l.policy.parameters()), torch.nn.utils.parameters_to_vector(l.value_net.parameters()))).clone()
    solo = {}
    for prec in ("fp32", "x3"):
        lr_, buf = make(prec)
        for _ in range(3):
            lr_.learn(buf)
        solo[prec] = vec(lr_)
    assert not torch.equal(solo["fp32"], solo["x3"])          # two different arithmetic paths ...
    assert relerr(solo["x3"], solo["fp32"]) < 1e-3             # ... for the same update
    (la, ba), (lb, bb) = make("fp32"), make("x3")
    for _ in range(3):
        la.learn(ba)
        lb.learn(bb)
    assert torch.equal(vec(la), solo["fp32"]) and torch.equal(vec(lb), solo["x3"])
    assert int(N.lib().rlppo_get_update_precision()) == 0
    # None = the process default: the same bits as "fp32" while the default is fp32
    lc, bc = make(None)
    for _ in range(3):
        lc.learn(bc)
    assert torch.equal(vec(lc), solo["fp32"])


def test_learn_single_call_report_and_magnitudes(golden):
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden("g5_learn_discrete")
    cfg = json.loads(str(g["cfg"]))
    learner = make_learner(cfg)
    buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    buf.submit_experience(*[g["exp." + k] for k in names])
    report = learner.learn(buf)
    for key in ("Cumulative Model Updates", "Policy Entropy", "Mean KL Divergence", "Value Function Loss",
                "SB3 Clip Fraction", "Policy Update Magnitude", "Value Function Update Magnitude"):
        ref = float(g["report." + key])
        assert abs(report[key] - ref) <= 5e-5 * max(abs(ref), 1e-4) + 1e-7, (key, report[key], ref)
    assert (learner.policy.arena.grad == 0).all()  # learn() leaves grads zeroed (ppo_learner.py:235-236)


def test_report_kernel_leaves_clean_sums_and_exact_magnitudes(golden):
    """[r6] rlppo_learn_report (the tail of learn() as one launch): the update magnitudes are the float64 norm of the float32 parameter
    differences (1e-6: the reference's own float32 norm is that far from it), the report sums are zeroed BY the kernel -- three
    consecutive learn() calls report what a learner whose sums are zeroed eagerly before every call reports, bit for bit -- and a
    learn() that raised half way leaves no stale sums behind."""
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden("g5_learn_discrete")
    cfg = json.loads(str(g["cfg"]))
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    runs = []
    for eager_zero in (False, True):
        learner = make_learner(cfg)
        buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
        buf.submit_experience(*[g["exp." + k] for k in names])
        reports = []
        for _ in range(3):
            if eager_zero:
                learner._stats_clean = False   # forces the fill at the start of learn()
            before = (learner.policy.arena.flat.clone(), learner.value_net.arena.flat.clone())
            r = learner.learn(buf)
            assert learner._stats_clean and float(learner._stats.abs().sum()) == 0.0
            for key, b, now in (("Policy Update Magnitude", before[0], learner.policy.arena.flat),
                                ("Value Function Update Magnitude", before[1], learner.value_net.arena.flat)):
                want = float((b - now).double().norm())
                assert want > 0 and abs(r[key] - want) <= 1e-6 * want, (key, r[key], want)
            reports.append({k: v for k, v in r.items() if k != "PPO Batch Consumption Time"})
        runs.append(reports)
    assert runs[0] == runs[1]
    # a learn() that dies between its passes and its report: the next call starts from clean sums all the same
    learner.grad_probe = lambda gr: (_ for _ in ()).throw(RuntimeError("probe"))
    with pytest.raises(RuntimeError, match="probe"):
        learner.learn(buf)
    assert not learner._stats_clean
    learner.grad_probe = None
    learner._grad_all.zero_()
    r = learner.learn(buf)
    assert 0.0 < r["Policy Entropy"] < np.log(cfg["n_act"]) + 1e-6


def test_buffer_too_small_reports_zeros(golden):
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden("g5_learn_discrete")
    cfg = json.loads(str(g["cfg"]))
    cfg["B"] = cfg["MB"] = 4096  # larger than the 1024 samples available (quirk Q7)
    learner = make_learner(cfg)
    buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    buf.submit_experience(*[g["exp." + k] for k in names])
    before = learner.policy.arena.flat.clone()
    r = learner.learn(buf)
    assert r["Policy Entropy"] == 0 and r["SB3 Clip Fraction"] == 0 and r["Policy Update Magnitude"] == 0
    assert torch.equal(before, learner.policy.arena.flat) and learner.cumulative_model_updates == 1


def test_experience_buffer_fifo_and_shuffle(golden):
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden("g8_fifo")
    for tag in ("under", "exact", "over", "huge", "stream"):
        buf = ExperienceBuffer(int(g[tag + ".size"]), 1, "cpu")
        base = 0
        for c in g[tag + ".chunks"]:
            ar = np.arange(base, base + c, dtype=np.float32)
            base += int(c)
            buf.submit_experience(ar[:, None].repeat(2, 1), ar, ar, ar, ar[:, None].repeat(2, 1), ar, ar, ar, ar)
        assert np.array_equal(buf.rewards.cpu().numpy(), g[tag + ".rewards"]), tag
        assert np.array_equal(buf.states.cpu().numpy(), g[tag + ".states"]), tag
    s = golden("g6_shuffle")
    buf = ExperienceBuffer(1000, 123, "cpu")
    ar = np.arange(1000, dtype=np.float32)
    sub = (ar[:, None].repeat(3, 1), ar, ar, ar, ar[:, None].repeat(3, 1), ar, ar, ar, ar)
    buf.submit_experience(*sub)
    for epoch in range(2):
        got = np.stack([b[0].cpu().numpy().astype(np.int64) for b in buf.get_all_batches_shuffled(300)])
        assert np.array_equal(got, s[f"epoch{epoch}"])  # bit-exact shuffle, remainder dropped
    buf.clear()
    buf.submit_experience(*sub)
    got = np.stack([b[0].cpu().numpy().astype(np.int64) for b in buf.get_all_batches_shuffled(300)])
    assert np.array_equal(got, s["after_clear"])


def test_policy_classes_get_action(golden):
    from rlgym_ppo_amd.ppo import ContinuousPolicy, DiscreteFF, MultiDiscreteFF, ValueEstimator
    g = golden("g1_discrete_forward")
    torch.manual_seed(11)
    pol = DiscreteFF(107, 90, (32, 32), "cuda:0")
    val = ValueEstimator(107, (32, 32), "cuda:0")
    for k, v in pol.state_dict().items():
        assert torch.equal(v.cpu(), torch.as_tensor(g["p." + k]))
    torch.manual_seed(999)
    st = torch.get_rng_state()
    q = torch.empty(64, 90).exponential_(1)
    torch.set_rng_state(st)
    act, logp = pol.get_action(g["obs"])  # draws its own noise from the CPU generator: same stream as q
    oact, ologp = nets.discrete_sample(torch.as_tensor(g["probs"]), q)
    assert act.dtype == torch.int64 and act.device.type == "cpu" and torch.equal(act, oact)
    assert np.abs(logp.numpy() - ologp.numpy()).max() < 1e-5
    act2, _ = pol.get_action(g["obs"], noise=g["q"])
    assert np.array_equal(act2.numpy(), g["actions"])
    assert relerr(pol.get_output(g["obs"]), torch.softmax(nets.mlp(nets.params_from_state(g, "p."), g["obs"]), -1)) < 1e-5
    det, lp0 = pol.get_action(g["obs"], deterministic=True)
    assert int(det) == int(g["det_action"]) and lp0 == 0
    g2 = golden("g2_value_forward")
    assert relerr(val(g2["obs"]), g2["values"]) < 1e-5 and val(g2["obs"]).shape == (64, 1)
    # state_dict round trip keeps the kernels in sync with the module
    sd = {k: v.clone() * 0.5 for k, v in pol.state_dict().items()}
    pol.load_state_dict(sd)
    half = [(w * 0.5, b * 0.5) for w, b in nets.params_from_state(g, "p.")]
    act3, _ = pol.get_action(g["obs"], noise=g["q"])
    assert torch.equal(act3, nets.discrete_sample(nets.discrete_probs(half, g["obs"]), torch.as_tensor(g["q"]))[0])

    c = golden("g9_continuous")
    torch.manual_seed(31)
    cp = ContinuousPolicy(231, 16, (48, 48), "cuda:0")
    a, lp = cp.get_action(c["obs"], noise=c["eps"])
    np.testing.assert_allclose(a.numpy(), c["act"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(lp.numpy(), c["logp"], rtol=2e-5, atol=2e-4)
    mean, std = cp.get_output(c["obs"])
    assert relerr(mean, c["mean"]) < 1e-5 and relerr(std, c["std"]) < 1e-5

    m = golden("g9_multidiscrete")
    torch.manual_seed(41)
    mp = MultiDiscreteFF(107, (32, 32), "cuda:0")
    a, lp = mp.get_action(m["obs"], noise=m["q"])
    assert np.array_equal(a.numpy(), m["act"])
    np.testing.assert_allclose(lp.numpy(), m["logp"], rtol=1e-5, atol=1e-5)
    det, _ = mp.get_action(m["obs"], deterministic=True)
    assert np.array_equal(det, m["det"])


def test_compute_gae_drop_in(golden):
    from rlgym_ppo_amd.util import torch_functions
    g = golden("g3_gae")
    p = "c5."
    vt, adv, ret = torch_functions.compute_gae(g[p + "rews"], g[p + "dones"], g[p + "trunc"], list(g[p + "values"]),
                                               gamma=0.99, lmbda=0.95, return_std=g[p + "ret_std"])
    assert vt.dtype == torch.float32 and vt.device.type == "cpu" and len(ret) == 512
    np.testing.assert_allclose(vt.numpy(), g[p + "value_targets"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv.numpy(), g[p + "advantages"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(np.asarray(ret), g[p + "returns"], rtol=1e-5, atol=1e-5)


def test_checkpoint_round_trip(tmp_path, golden):
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden("g5_learn_discrete")
    meta = json.loads(str(golden("g10_checkpoint")["meta"]))
    cfg = json.loads(str(g["cfg"]))
    learner = make_learner(cfg)
    buf = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    buf.submit_experience(*[g["exp." + k] for k in names])
    learner.n_epochs = 1
    learner.learn(buf)
    learner.save_to(str(tmp_path))
    assert list(learner.policy.state_dict().keys()) == meta["policy_keys"]
    assert list(learner.value_net.state_dict().keys()) == meta["value_keys"]
    osd = torch.load(str(tmp_path / "PPO_POLICY_OPTIMIZER.pt"))
    assert sorted(osd["state"][0].keys()) == meta["adam_state_keys"] and sorted(osd["state"].keys()) == meta["adam_state_ids"]
    assert sorted(osd["param_groups"][0].keys()) == meta["adam_param_group_keys"]
    # a second learner resumes from the files and then tracks the first one exactly
    other = make_learner(cfg)
    other.n_epochs = 1
    other.load_from(str(tmp_path))
    buf2 = ExperienceBuffer(cfg["n"], cfg["seed"], "cpu")
    buf2.submit_experience(*[g["exp." + k] for k in names])
    buf2.epoch_indices()  # first learner consumed one permutation already
    learner.learn(buf)
    other.learn(buf2)
    assert relerr(other.policy.arena.flat, learner.policy.arena.flat) < 1e-6
    assert relerr(other.value_net.arena.flat, learner.value_net.arena.flat) < 1e-6
    # and the reference's own optimiser class can read the file
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p)) for p in learner.policy.parameters()], lr=1.0)
    ref_opt.load_state_dict(osd)


def test_shuffle_pipeline_is_transparent_on_the_device(golden, monkeypatch):
    """The look-ahead shuffle pipeline (helper threads, own upload stream, ring of index vectors) must not change anything
    observable: device index vectors == numpy's stream epoch by epoch, and learn() with look-ahead 3 ends in bit-identical
    parameters to learn() with look-ahead 0, including across a buffer that grows between learn() calls (the speculation is
    dropped) and a host-side epoch_indices() call in between."""
    from rlgym_ppo_amd.ppo import ExperienceBuffer
    g = golden("g5_learn_discrete")
    cfg = json.loads(str(g["cfg"]))
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    exp = [g["exp." + k] for k in names]

    # 1. the device vectors are numpy's stream; a vector stays intact while later epochs are drawn and uploaded
    monkeypatch.setenv("RLPPO_SHUFFLE_LOOKAHEAD", "3")
    buf = ExperienceBuffer(1 << 20, 7, "cpu")
    n = 300000
    z = torch.zeros(n, device="cuda")
    buf.submit_experience(torch.zeros(n, 4, device="cuda"), z, z, z, torch.zeros(n, 4, device="cuda"), z, z, z, z)
    ref = np.random.RandomState(7)
    for epoch in range(12):
        dev = buf.epoch_indices_device()
        got = dev.cpu().numpy()
        assert np.array_equal(got, ref.permutation(n)), epoch
    assert np.array_equal(buf.epoch_indices(), ref.permutation(n))       # host view of the same stream
    assert buf.rng.randint(1 << 30) == ref.randint(1 << 30)              # foreign draw: speculation dropped
    assert np.array_equal(buf.epoch_indices_device().cpu().numpy(), ref.permutation(n))

    # 2. learn() is bit-identical with and without look-ahead
    finals = []
    for lookahead in ("0", "3"):
        monkeypatch.setenv("RLPPO_SHUFFLE_LOOKAHEAD", lookahead)
        learner = make_learner(cfg)
        learner.n_epochs = 5
        b = ExperienceBuffer(4 * cfg["n"], cfg["seed"], "cpu")
        b.submit_experience(*exp)
        learner.learn(b)
        learner.learn(b)
        b.submit_experience(*exp)                                        # the buffer grows: n changes
        learner.learn(b)
        b.epoch_indices()                                                # somebody else consumes an epoch
        learner.learn(b)
        torch.cuda.synchronize()
        finals.append((learner.policy.arena.flat.clone(), learner.value_net.arena.flat.clone(), b.rng.get_state()[1].copy()))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])
    assert np.array_equal(finals[0][2], finals[1][2])


def test_fused_minibatches_equal_separate_passes():
    """PPOLearner evaluates consecutive minibatches of a batch in one pass (max_fused_minibatches): the reference sums the
    MB/B-scaled minibatch gradients before one clip + Adam (ppo_learner.py:134-193), so the fused pass must give the same
    gradient norm, the same parameters and the same report as eight separate passes, up to fp32 summation order."""
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    n, d, A, B, MB = 65536, 107, 90, 65536, 8192
    rs = np.random.RandomState(3)
    obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    res = []
    for fused in (1, 8):
        torch.manual_seed(3)
        learner = PPOLearner(d, A, 0, (256, 256, 256), (256, 256, 256), (0.1, 1.0), B, 3, 3e-4, 3e-4, 0.2, 0.005, MB, "cuda:0")
        learner.max_fused_minibatches = fused
        if fused == 1:
            noise = torch.as_tensor(np.random.RandomState(4).exponential(size=(n, A)).astype(np.float32))
            act, logp = learner.policy.get_action(obs, noise=noise)
            exp = (obs, act.numpy().astype(np.float32), logp.numpy() + 0.05 * rs.randn(n).astype(np.float32), np.zeros(n, np.float32),
                   obs, np.zeros(n, np.float32), np.zeros(n, np.float32), rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32))
        buf = ExperienceBuffer(n, 9, "cpu")
        buf.submit_experience(*exp)
        p0, v0 = learner.policy.arena.flat.clone(), learner.value_net.arena.flat.clone()
        learner.n_epochs = 1
        learner.learn(buf)                      # first optimiser step: identical parameters going in
        gp, gv = learner.policy_optimizer.gnorm2.item(), learner.value_optimizer.gnorm2.item()
        learner.n_epochs = 2
        report = learner.learn(buf)
        torch.cuda.synchronize()
        res.append((learner.policy.arena.flat.clone(), learner.value_net.arena.flat.clone(), report, gp, gv))
    (p1, v1, r1, gp1, gv1), (p8, v8, r8, gp8, gv8) = res
    assert abs(gp8 - gp1) <= 1e-5 * gp1 and abs(gv8 - gv1) <= 1e-5 * gv1          # squared gradient norms of the first step
    # Adam's first steps are sign-like (m/sqrt(v) = g/|g|): an element whose gradient is within summation noise of zero can
    # move by up to 2 lr either way, so the parameters are compared through the L2 distance of the three-step updates (a lost
    # or double-counted slice changes the update by O(1) of its norm)
    for a, b, o in ((p8, p1, p0), (v8, v1, v0)):
        assert ((a - b).double().norm() / (b - o).double().norm()).item() < 2e-3
        assert relerr(a, b) < 3 * 2 * 3e-4 / b.abs().max().item()
    for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction", "Policy Update Magnitude",
              "Value Function Update Magnitude"):
        assert abs(r8[k] - r1[k]) <= 5e-5 * max(abs(r1[k]), 1e-3) + 1e-7, (k, r8[k], r1[k])
    assert r8["Cumulative Model Updates"] == r1["Cumulative Model Updates"] == 3


def test_get_action_graph_replay_equals_eager_path():
    """DiscreteFF.get_action replays one hipGraph per batch-size bucket (copies + staging + forward + sampling + read-back):
    same actions and log-probabilities, bit for bit, as the eager path; the graph reads the CURRENT weights (the packed copy
    is refreshed in place after an optimiser step), serves every n of its bucket and float64 observations."""
    from rlgym_ppo_amd.ppo import DiscreteFF
    torch.manual_seed(21)
    pol = DiscreteFF(107, 90, (64, 64), "cuda:0")
    rs = np.random.RandomState(2)
    for round_ in range(2):
        for n in (1, 8, 17, 80, 129, 700):
            obs = np.clip(rs.randn(n, 107), -5, 5).astype(np.float32 if n % 2 else np.float64)
            q = torch.empty(n, 90).exponential_(1)
            pol.act_graphs = True
            a1, l1 = pol.get_action(obs, noise=q)
            pol.act_graphs = False
            a0, l0 = pol.get_action(obs, noise=q)
            assert a1.dtype == torch.int64 and a1.device.type == "cpu" and a1.shape == (n,)
            assert torch.equal(a0, a1) and torch.equal(l0, l1), (round_, n)
        # default noise: the same CPU generator stream either way
        obs = np.clip(rs.randn(33, 107), -5, 5).astype(np.float32)
        pol.act_graphs = True
        torch.manual_seed(5)
        a1, l1 = pol.get_action(obs)
        pol.act_graphs = False
        torch.manual_seed(5)
        a0, l0 = pol.get_action(obs)
        assert torch.equal(a0, a1) and torch.equal(l0, l1)
        with torch.no_grad():  # new weights, written THROUGH the Parameters (what a stock optimiser / vector_to_parameters does:
            for p in pol.parameters():  # p._version moves, flat._version does not): the next round must see them
                p.add_(torch.randn_like(p) * 0.05)
    assert set(pol._graphs) == {16, 32, 80, 256, 1024, 48}
    pol.act_graphs = True
    big = np.clip(rs.randn(1500, 107), -5, 5).astype(np.float32)      # above act_graph_max: the eager path, no new graph
    pol.get_action(big)
    assert set(pol._graphs) == {16, 32, 80, 256, 1024, 48}


def test_get_action_draws_its_noise_behind_the_launch(monkeypatch):
    """[r5] The graph-served get_action of the discrete head writes its observations into a host window (device memory behind the
    PCIe aperture), launches, and draws its Exp(1) noise afterwards (ActGraph.push / .late, rlppo_act_opts.noise_ctl): same
    actions, log-probabilities and generator state as the form that keeps everything in pinned host memory and stages the noise
    before the launch (RLPPO_ACT_PUSH=0: [r6] the one fallback left), call after call with changing n inside one bucket; a draw
    that raises leaves no kernel waiting and the next call is served normally; a host that is held up gets a second launch."""
    from rlgym_ppo_amd.ppo import DiscreteFF
    torch.manual_seed(3)
    pol = DiscreteFF(107, 90, (256, 256, 256), "cuda:0")
    rs = np.random.RandomState(4)
    calls = [np.clip(rs.randn(n, 107), -5, 5).astype(np.float32) for n in (8, 3, 16, 80, 70, 80, 1, 33, 128, 200, 700)]
    out = {}
    for push in ("1", "0"):
        monkeypatch.setenv("RLPPO_ACT_PUSH", push)
        pol._graphs.clear()
        torch.manual_seed(99)
        out[push] = [pol.get_action(o) for o in calls] + [torch.get_rng_state()]
        assert all(g.push == (push == "1") and g.late == (push == "1" and g.cap <= 256) for g in pol._graphs.values())
        assert set(pol._graphs) == {16, 48, 80, 128, 256, 1024}
        # (poll timeouts / second launches are allowed -- a busy host may hold a call up -- and change nothing: the results below decide)
    for (a1, l1), (a0, l0) in zip(out["1"][:-1], out["0"][:-1]):
        assert torch.equal(a1, a0) and torch.equal(l1, l0)
    assert torch.equal(out["1"][-1], out["0"][-1])
    monkeypatch.setenv("RLPPO_ACT_PUSH", "1")
    pol._graphs.clear()
    good = pol._draw_noise

    def broken(n):
        raise KeyError("no noise today")

    torch.manual_seed(99)
    a_first = pol.get_action(calls[0])
    pol._draw_noise = broken
    with pytest.raises(KeyError):
        pol.get_action(calls[0])
    torch.cuda.synchronize()
    pol._draw_noise = good
    torch.manual_seed(99)
    a_again = pol.get_action(calls[0])
    assert torch.equal(a_first[0], a_again[0]) and torch.equal(a_first[1], a_again[1])
    # a host that is held up between the launch and the publish for longer than the kernel's patience (20 ms): the kernel gets
    # out of the way, the call is made again with the noise in place -- same results, one retry counted
    import time

    def slow(n):
        time.sleep(0.05)
        return good(n)

    pol._draw_noise = slow
    torch.manual_seed(99)
    a_slow = pol.get_action(calls[0])
    pol._draw_noise = good
    g = pol._graphs[16]
    assert g.late_retries >= 1 and torch.equal(a_first[0], a_slow[0]) and torch.equal(a_first[1], a_slow[1])
    # parameters written behind the packed copy's back (a stock optimiser's in-place step): the launch goes out on the stale copy,
    # the check behind it notices, the call is made again on the new weights -- exactly what the eager path computes
    before = g.stale_relaunches
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    q = torch.empty(8, 90).exponential_(1)
    a_new = pol.get_action(calls[0], noise=q)
    assert g.stale_relaunches == before + 1
    pol.act_graphs = False
    a_ref = pol.get_action(calls[0], noise=q)
    pol.act_graphs = True
    assert torch.equal(a_new[0], a_ref[0]) and torch.equal(a_new[1], a_ref[1])
    assert not torch.equal(a_new[1], a_first[1])
    a_new2 = pol.get_action(calls[0], noise=q)
    assert g.stale_relaunches == before + 1 and torch.equal(a_new2[1], a_ref[1])


@pytest.mark.parametrize("name,hidden", [("g1b_discrete_forward_128x2", (128, 128)), ("g1c_discrete_forward_256x3", (256, 256, 256))])
def test_g1bc_through_the_collectors_small_call(golden, name, hidden):
    """[r5] The reference-held vectors G1b / G1c through DiscreteFF.get_action AS THE COLLECTOR CALLS IT -- host observations, a graph
    replay on a host window, the noise published behind the launch, completion words polled (ActGraph.push / .late): action indices
    EXACT, log-probabilities within 1e-5, in chunks of the reference's call sizes (8 ... 80 observations) and whole; the library's
    counter says the one-launch kernel was what the graphs captured."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import DiscreteFF
    g = golden(name)
    assert float(g["margin"].min()) > 1e-4
    pol = DiscreteFF(107, 90, hidden, "cuda:0")
    pol.load_state_dict({k[2:]: torch.as_tensor(g[k]) for k in list(g.keys()) if k.startswith("p.")})
    obs, q = np.asarray(g["obs"], dtype=np.float32), torch.as_tensor(np.asarray(g["q"], dtype=np.float32))
    n = obs.shape[0]
    fused0 = int(N.lib().rlppo_dbg_counter(0))
    start = 0
    for size in (8, 16, 3, 80, 33, n):
        lo, hi = (0, n) if size == n else (start % n, min(start % n + size, n))
        start += size
        a, lp = pol.get_action(obs[lo:hi], noise=q[lo:hi])
        assert np.array_equal(a.numpy(), g["actions"][lo:hi]), (name, lo, hi)
        assert np.abs(lp.numpy() - g["logp"][lo:hi]).max() < 1e-5
    assert pol._graphs and all(gr.push and gr.late == (gr.cap <= 256) for gr in pol._graphs.values())
    assert int(N.lib().rlppo_dbg_counter(0)) > fused0


@pytest.mark.parametrize("head", ["discrete", "gaussian", "multidiscrete"])
def test_small_call_fallbacks_all_give_the_same_bits(monkeypatch, head):
    """[r6] The small get_action call has ONE configuration (host window, late noise where the head has the one-launch step, polled
    completion words) and these ways of NOT using it, every one the same actions, log-probabilities and generator state, for every
    head, at 1 / 8 / 80 / 200 observations: RLPPO_ACT_PUSH=0 (the documented fallback: pinned host memory); a device that refuses a
    host window (rlppo_dbg_set(41, 1) answers like one without a large BAR: the graphs fall back to pinned memory by themselves); a
    device whose windows have no flush register (41, 2: rlppo_host_window_flush does nothing); the general path (policy.act_graphs =
    False: what calls above 1024 rows take); RLPPO_TUNE-style 128-row tiles in the layer chain (40, 0)."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ContinuousPolicy, DiscreteFF, MultiDiscreteFF
    torch.manual_seed(17)
    if head == "discrete":
        pol, d = DiscreteFF(107, 90, (256, 256, 256), "cuda:0"), 107
    elif head == "gaussian":
        pol, d = ContinuousPolicy(61, 16, (512, 512), "cuda:0"), 61
    else:
        pol, d = MultiDiscreteFF(107, (256, 256), "cuda:0"), 107
    rs = np.random.RandomState(6)
    calls = [np.clip(rs.randn(n, d), -5, 5).astype(np.float32) for n in (1, 8, 80, 200, 8)]

    def run(env=None, tune=None, graphs=True, want_push=None):
        monkeypatch.delenv("RLPPO_ACT_PUSH", raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        pol._graphs.clear()
        pol.act_graphs = graphs
        if tune:
            N.check(N.lib().rlppo_dbg_set(*tune[0]))
        try:
            torch.manual_seed(4321)
            out = [pol.get_action(o) for o in calls]
            if want_push is not None:
                assert pol._graphs and all(g.push == want_push for g in pol._graphs.values()), (head, env, tune)
            return out, torch.get_rng_state()
        finally:
            if tune:
                N.check(N.lib().rlppo_dbg_set(*tune[1]))
            pol.act_graphs = True
            pol._graphs.clear()

    ref, ref_state = run(want_push=True)
    for kw in (dict(env={"RLPPO_ACT_PUSH": "0"}, want_push=False), dict(tune=((41, 1), (41, 0)), want_push=False),
               dict(tune=((41, 2), (41, 0)), want_push=True), dict(graphs=False), dict(tune=((40, 0), (40, 1)))):
        out, state = run(**kw)
        for (a1, l1), (a0, l0) in zip(out, ref):
            assert torch.equal(torch.as_tensor(a1), torch.as_tensor(a0)) and torch.equal(torch.as_tensor(l1), torch.as_tensor(l0)), (head, kw)
        assert torch.equal(state, ref_state), (head, kw)


@pytest.mark.parametrize("n_agents,steps", [(64, 128), (768, 24)], ids=["on_the_spot", "look_ahead"])
def test_seeded_rollout_draws_the_reference_noise_stream(n_agents, steps):
    """Rollout steps from one seed: DiscreteFF.get_action's default noise (librlppo's host implementation of torch's CPU
    exponential_ -- engine.HostExponential) is the stream the reference's torch.multinomial consumes, so the action indices are
    the oracle's (identical noise; an index may only differ on a near-tie of p/q caused by the ulp-level difference of the
    probabilities: none expected, margin stated) and the generator ends in the reference's state.  64 agents x 90 actions: the
    draw happens on the spot (below HostExponential.LOOKAHEAD_MIN numbers the hand-over to the helper thread costs more than
    the draw); 768 agents: drawn one step ahead on the helper thread -- every step but the first is served from it."""
    from rlgym_ppo_amd import engine
    from rlgym_ppo_amd.ppo import DiscreteFF
    torch.manual_seed(77)
    pol = DiscreteFF(107, 90, (64, 64), "cuda:0")
    params = [(l.weight.detach().cpu(), l.bias.detach().cpu()) for l in pol.arena.linears]
    rs = np.random.RandomState(5)
    obs = [np.clip(rs.randn(n_agents, 107), -5, 5).astype(np.float32) for _ in range(steps)]
    torch.manual_seed(1234)
    want = []
    for o in obs:
        q = nets.draw_exp_noise(n_agents, 90)
        p = nets.discrete_probs(params, o)
        want.append((nets.discrete_sample(p, q), p, q))
    s_ref = torch.get_rng_state()
    ahead = n_agents * 90 >= engine.HostExponential.LOOKAHEAD_MIN
    for graphs in (True, False):
        pol.act_graphs = graphs
        torch.manual_seed(1234)
        engine._HOST_EXP = None
        mismatches = 0
        for o, ((a_ref, lp_ref), p, q) in zip(obs, want):
            a, lp = pol.get_action(o)
            bad = (a != a_ref).nonzero().flatten().tolist()
            for i in bad:
                s = p[i] / q[i]
                assert abs(s[a[i]] - s[a_ref[i]]) <= 1e-5 * s[a_ref[i]], "index mismatch that is not a near-tie"
            mismatches += len(bad)
            same = a == a_ref
            assert (lp[same] - lp_ref[same]).abs().max().item() < 1e-5
        assert mismatches <= 1
        assert torch.equal(torch.get_rng_state(), s_ref)
        if ahead:
            assert engine._HOST_EXP.hits >= steps - 2           # every step but the first was served from the look-ahead
        else:
            assert engine._HOST_EXP.hits == 0 and engine._HOST_EXP.misses == steps


def test_captured_act_graphs_follow_the_inference_precision():
    """A captured graph replays the kernels selected at capture time; the graph cache is keyed on the library's selection
    epoch, so set_inference_precision takes effect for small (graph-served, n <= 1024) batches too, both ways."""
    from rlgym_ppo_amd.engine import set_inference_precision
    from rlgym_ppo_amd.ppo import ContinuousPolicy
    torch.manual_seed(4)
    pol = ContinuousPolicy(231, 16, (512, 512), "cuda:0")
    rs = np.random.RandomState(4)
    obs = np.clip(rs.randn(200, 231), -5, 5).astype(np.float32)
    eps = torch.as_tensor(rs.randn(200, 8).astype(np.float32))
    a32, _ = pol.get_action(obs, noise=eps)            # captures the fp32 graph of bucket 256
    assert 256 in pol._graphs
    pol.act_graphs = False
    e32, _ = pol.get_action(obs, noise=eps)
    set_inference_precision("bf16")
    try:
        e16, _ = pol.get_action(obs, noise=eps)        # eager bf16
        pol.act_graphs = True
        a16, _ = pol.get_action(obs, noise=eps)        # must NOT replay the fp32 graph
    finally:
        set_inference_precision("fp32")
    a_back, _ = pol.get_action(obs, noise=eps)
    assert torch.equal(a32, e32) and torch.equal(a16, e16) and torch.equal(a_back, a32)
    assert not torch.equal(a16, a32) and (a16 - a32).abs().max().item() < 0.1


def test_get_action_graph_replay_other_heads():
    """The graph-replayed rollout step of the Gaussian and the multi-discrete heads: bit-identical to their eager paths, with
    given noise and with the default CPU-generator draw."""
    from rlgym_ppo_amd.ppo import ContinuousPolicy, MultiDiscreteFF
    torch.manual_seed(22)
    rs = np.random.RandomState(3)
    heads = [(ContinuousPolicy(231, 16, (64, 64), "cuda:0"), 231, lambda n: torch.empty(n, 8).normal_(0, 1)),
             (MultiDiscreteFF(107, (64, 64), "cuda:0"), 107, lambda n: torch.empty(n * 8, 3).exponential_(1))]
    for pol, d, draw in heads:
        for n in (3, 40, 300):
            obs = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
            q = draw(n)
            pol.act_graphs = True
            a1, l1 = pol.get_action(obs, noise=q)
            pol.act_graphs = False
            a0, l0 = pol.get_action(obs, noise=q)
            assert a1.shape == a0.shape and a1.dtype == a0.dtype and torch.equal(a0, a1) and torch.equal(l0, l1), (type(pol).__name__, n)
            pol.act_graphs = True
            torch.manual_seed(9)
            a1, l1 = pol.get_action(obs)
            pol.act_graphs = False
            torch.manual_seed(9)
            a0, l0 = pol.get_action(obs)
            assert torch.equal(a0, a1) and torch.equal(l0, l1)
        assert set(pol._graphs) == {16, 48, 512}


def test_discrete_step_host_and_device_outputs():
    """DiscreteFF.step [r3]: the same step with its results in pinned host memory (to_host=True: what get_action returns), with only the
    action indices on the host (to_host="actions": the vectorised rollout) and on the device -- identical numbers, and identical to
    get_action's; the optional device destinations (padded rows, float actions, log-probs) are filled."""
    torch.manual_seed(3)
    from rlgym_ppo_amd.ppo import DiscreteFF
    pol = DiscreteFF(107, 90, (256, 256, 256), "cuda:0")
    rs = np.random.RandomState(3)
    obs = np.clip(rs.randn(300, 107), -5, 5).astype(np.float32)
    q = torch.as_tensor(rs.exponential(size=(300, 90)).astype(np.float32))
    a0, lp0 = pol.get_action(obs, noise=q)                      # ActGraph (n <= 1024, host input)
    a1, lp1 = pol.step(obs, noise=q)                            # fused launch, pinned outputs
    rows = torch.empty(300, pol.arena.ld_in, device="cuda")
    af, lpd = torch.empty(300, device="cuda"), torch.empty(300, device="cuda")
    a2, lp2 = pol.step(torch.from_numpy(obs).cuda(), noise=q.cuda(), rows_out=rows, actions_f32=af, logp_out=lpd, to_host="actions")
    a3, lp3 = pol.step(obs, noise=q, to_host=False)
    assert not a1.is_cuda and not a2.is_cuda and lp2.is_cuda and a3.is_cuda
    for a, lp in ((a1, lp1), (a2, lp2), (a3, lp3)):
        assert torch.equal(a.cpu(), a0) and torch.equal(lp.cpu(), lp0)
    assert lp2.data_ptr() == lpd.data_ptr() and torch.equal(af.cpu(), a0.float())
    assert torch.equal(rows, pol.arena.stage_obs(obs))
    # standardisation fused in: the reference's scalars
    b0, _ = pol.get_action(obs, noise=q, standardize=(0.25, 2.0))
    b1, _ = pol.step(obs, noise=q, standardize=(0.25, 2.0))
    assert torch.equal(b0, b1)


@pytest.mark.gpu
def test_learn_recovers_from_an_optimiser_barrier_that_gives_up():
    """PPOLearner.learn with the one-launch optimiser tail when its grid barrier gives up (forced with the library's test hooks):
    the call raises OptimizerBarrierTimeout, parameters and Adam moments are untouched and finite (no NaN, nothing half-applied),
    gradients are zeroed, the learner has switched to the three-operation tail, and the NEXT learn() on the same buffer gives
    exactly what a learner that never met the failure gives."""
    from rlgym_ppo_amd import _native as N
    from rlgym_ppo_amd.ppo import ExperienceBuffer, PPOLearner
    from rlgym_ppo_amd.ppo.ppo_learner import OptimizerBarrierTimeout
    L = N.lib()

    def make():
        torch.manual_seed(11)
        np.random.seed(11)
        lr = PPOLearner(20, 6, 0, (64, 64), (64, 64), (0.1, 1.0), 512, 2, 3e-4, 3e-4, 0.2, 0.005, 256, "cuda:0")
        rs = np.random.RandomState(3)
        obs = rs.randn(1024, 20).astype(np.float32)
        act, logp = lr.policy.get_action(obs, noise=torch.empty(1024, 6).exponential_(1, generator=torch.Generator().manual_seed(5)).cuda())
        buf = ExperienceBuffer(1024, 7, "cpu")
        z = np.zeros(1024, np.float32)
        buf.submit_experience(obs, act.numpy().astype(np.float32), logp.numpy(), z, obs, z, z, rs.randn(1024).astype(np.float32), rs.randn(1024).astype(np.float32))
        return lr, buf

    a, buf_a = make()
    b, buf_b = make()
    flat = lambda lr: torch.cat([torch.nn.utils.parameters_to_vector(lr.policy.parameters()), torch.nn.utils.parameters_to_vector(lr.value_net.parameters())]).clone()
    p0 = flat(a)
    assert a.one_launch_optimizer
    N.check(L.rlppo_dbg_set(34, 0))
    N.check(L.rlppo_dbg_set(35, 1))
    try:
        with pytest.raises(OptimizerBarrierTimeout):
            a.learn(buf_a)
    finally:
        N.check(L.rlppo_dbg_set(35, 0))
        N.check(L.rlppo_dbg_set(34, -1))
    assert torch.equal(flat(a), p0) and torch.isfinite(flat(a)).all()
    assert not a.one_launch_optimizer and int(a._grad_all.abs().sum().item()) == 0 and int(a._opt_sync.abs().sum().item()) == 0
    # the failed call consumed the buffer's permutations as a completed one would have (same generator on both learners)
    b.one_launch_optimizer = False
    buf_b.epoch_indices(); buf_b.epoch_indices()
    ra, rb = a.learn(buf_a), b.learn(buf_b)
    assert torch.equal(flat(a), flat(b))
    for k in ("Policy Entropy", "Mean KL Divergence", "Value Function Loss", "SB3 Clip Fraction"):
        assert ra[k] == rb[k], k
