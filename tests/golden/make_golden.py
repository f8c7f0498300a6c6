#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference implementation (AechPro/rlgym-ppo @ v1.3.13) with `wandb`/`gym` stubbed
(SURVEY.md section 8(c)), drives its hot-path functions on small seeded inputs, and stores the
inputs + outputs as .npz fixtures next to this script.  The fixtures are DATA: no reference source
or bytecode is written anywhere.  Nothing in tests/, bench.py or smoke() imports the reference;
they read only the .npz files produced here.

    python tests/golden/make_golden.py          # regenerates every fixture

Versions that produced the committed fixtures are stored inside each file (key `_versions`).
Fixture ids follow SURVEY.md section 8(c): G1..G10.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    wandb = types.ModuleType("wandb")
    wr = types.ModuleType("wandb.wandb_run")
    wr.Run = object
    wandb.wandb_run = wr
    sys.modules["wandb"] = wandb
    sys.modules["wandb.wandb_run"] = wr
    sys.modules["gym"] = types.ModuleType("gym")
    sys.path.insert(0, REF)
    import rlgym_ppo  # noqa: F401
    from rlgym_ppo.ppo import (ContinuousPolicy, DiscreteFF, ExperienceBuffer, MultiDiscreteFF,
                               PPOLearner, ValueEstimator)
    from rlgym_ppo.util import WelfordRunningStat, torch_functions
    return dict(ContinuousPolicy=ContinuousPolicy, DiscreteFF=DiscreteFF, ExperienceBuffer=ExperienceBuffer,
                MultiDiscreteFF=MultiDiscreteFF, PPOLearner=PPOLearner, ValueEstimator=ValueEstimator,
                WelfordRunningStat=WelfordRunningStat, torch_functions=torch_functions)


VERSIONS = json.dumps({"numpy": np.__version__, "torch": torch.__version__, "reference": "rlgym-ppo 1.3.13"})


def save(name, **arrays):
    arrays["_versions"] = np.array(VERSIONS)
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: (v.shape, str(v.dtype)) for k, v in out.items() if not k.startswith("_")})


def params_of(module, prefix):
    return {f"{prefix}{k}": v.detach().clone() for k, v in module.state_dict().items()}


def grads_of(module, prefix):
    return {f"{prefix}{k}": p.grad.detach().clone() for k, p in module.named_parameters()}


# ---------------------------------------------------------------------------------------------
def g1_g2_forward(R):
    """G1: DiscreteFF.get_action (discrete_policy.py:44-62) incl. the Exp(1) noise; G2: ValueEstimator."""
    torch.manual_seed(11)
    pol = R["DiscreteFF"](107, 90, (32, 32), "cpu")
    val = R["ValueEstimator"](107, (32, 32), "cpu")
    rs = np.random.RandomState(5)
    obs = np.clip(rs.randn(64, 107), -5, 5).astype(np.float32)
    with torch.no_grad():
        probs = torch.clamp(pol.get_output(obs).view(-1, 90), min=1e-11, max=1)
        st = torch.get_rng_state()
        actions, logp = pol.get_action(obs)
        torch.set_rng_state(st)
        q = torch.empty(64, 90).exponential_(1)  # same draw multinomial consumed
        det_action, det_lp = pol.get_action(obs, deterministic=True)
        values = val(obs)
        # float64 observations are coerced to f32 by the reference (value_estimator.py:30-36)
        values64 = val(obs.astype(np.float64))
    save("g1_discrete_forward", obs=obs, q=q, probs=probs, actions=actions, logp=logp,
         det_action=np.int64(det_action), **params_of(pol, "p."))
    save("g2_value_forward", obs=obs, values=values, values_from_f64=values64, **params_of(val, "v."))


def g1bc_forward_fused_shapes(R):
    """G1b / G1c [r4]: DiscreteFF.get_action (discrete_policy.py:44-62) on network shapes the ONE-LAUNCH rollout kernel covers
    (csrc/fused_act.hip: hidden widths 64 / 128 / 256; G1's 32-wide nets only ever pass through the layer chain): (128, 128) -- with 107 observations the first padded width, 112, must fit the hidden width -- and the
    BASELINE configs[1] shape (256, 256, 256), 64 rows each, the Exp(1) noise recorded as in G1."""
    for tag, layers, seed in (("g1b_discrete_forward_128x2", (128, 128), 12), ("g1c_discrete_forward_256x3", (256, 256, 256), 13)):
        torch.manual_seed(seed)
        pol = R["DiscreteFF"](107, 90, layers, "cpu")
        rs = np.random.RandomState(seed)
        obs = np.clip(rs.randn(64, 107), -5, 5).astype(np.float32)
        with torch.no_grad():
            probs = torch.clamp(pol.get_output(obs).view(-1, 90), min=1e-11, max=1)
            st = torch.get_rng_state()
            actions, logp = pol.get_action(obs)
            torch.set_rng_state(st)
            q = torch.empty(64, 90).exponential_(1)
        # margin of every row's selection: the two largest p/q (a near-tie may legitimately go the other way in another fp32 forward)
        ratio = (probs / q).numpy().astype(np.float64)
        top2 = np.sort(ratio, axis=1)[:, -2:]
        save(tag, obs=obs, q=q, probs=probs, actions=actions, logp=logp, margin=(top2[:, 1] - top2[:, 0]) / top2[:, 1],
             layers=np.asarray(layers), **params_of(pol, "p."))


def param_hash(vec):
    """Order-sensitive exact hash of a float32 vector's BITS (uint64 wrap-around arithmetic: the same on every host)."""
    bits = np.ascontiguousarray(np.asarray(vec, np.float32)).view(np.uint32).astype(np.uint64)
    mult = (np.arange(bits.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    with np.errstate(over="ignore"):
        return np.bitwise_xor.reduce(bits * mult) ^ np.uint64(bits.size)


def g5big_inputs(cfg):
    """The experience of G5big from seeds alone (numpy legacy generators: the same bits on every host), so that a test can rebuild
    it where the fixture cannot hold it (262,144 x 107 states = 112 MB).  Actions are uniform, the 'old' log-probabilities a
    noisy log(1/A): ratios spread over both clip edges from the first step on.  Returned in submit_experience order."""
    rs = np.random.RandomState(cfg["seed"] + 1)
    n, d, A = cfg["n"], cfg["d"], cfg["n_act"]
    states = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    actions = rs.randint(0, A, n).astype(np.float32)
    log_probs = (np.log(1.0 / A) + 0.15 * rs.randn(n)).astype(np.float32)
    rewards = rs.randn(n).astype(np.float32)
    dones = (rs.rand(n) < 0.03).astype(np.float32)
    trunc = ((rs.rand(n) < 0.03) & (dones == 0)).astype(np.float32)
    values = rs.randn(n).astype(np.float32)
    adv = rs.randn(n).astype(np.float32)
    return states, actions, log_probs, rewards, states[:1].repeat(n, 0), dones, trunc, values, adv


def g5big_learn(R):
    """G5big [r4]: PPOLearner.learn (ppo_learner.py:92-238) at the size where the product's paired / gather-fused launches engage
    (csrc/api.hip: from 262,144 rows per pass): 256x3 nets, n = B = 262,144, MB = 65,536, 2 epochs = 2 optimiser steps of 4
    minibatches.  The inputs come from seeds (g5big_inputs: a test rebuilds them); stored: an exact bit hash + the first 64 values
    of the initial parameters, every 8th entry + norms of the first step's batch gradient (before clipping), the first 64 values +
    plain sums of the parameters after step 0, the FULL parameter vectors after the last step, the report.  ~2 minutes on one thread."""
    cfg = dict(policy_type=0, d=107, n_act=90, layers=(256, 256, 256), n=262144, B=262144, MB=65536, epochs=2, seed=321,
               lr=3e-4, clip=0.2, ent=0.005)
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    learner = R["PPOLearner"](cfg["d"], cfg["n_act"], 0, cfg["layers"], cfg["layers"], (0.1, 1.0), cfg["B"], cfg["epochs"], cfg["lr"],
                              cfg["lr"], cfg["clip"], cfg["ent"], cfg["MB"], "cpu")
    vec = lambda m: torch.nn.utils.parameters_to_vector(m.parameters()).detach().clone().numpy()
    out = {"p0.hash": param_hash(vec(learner.policy)), "v0.hash": param_hash(vec(learner.value_net)),
           "p0.head": vec(learner.policy)[:64], "v0.head": vec(learner.value_net)[:64]}
    buf = R["ExperienceBuffer"](cfg["n"], cfg["seed"], "cpu")
    exp = g5big_inputs(cfg)
    out["exp.hash"] = np.asarray([param_hash(x.reshape(-1)) for x in exp[:3] + exp[7:]])
    buf.submit_experience(*exp)
    step_no = [0]
    orig = learner.value_optimizer.step

    def rec(*a, **k):  # the value optimiser steps last (ppo_learner.py:192-193)
        r = orig(*a, **k)
        s = step_no[0]
        p, v = vec(learner.policy), vec(learner.value_net)
        out[f"step{s}.policy_head"], out[f"step{s}.value_head"] = p[:64], v[:64]
        out[f"step{s}.policy_sum"], out[f"step{s}.value_sum"] = np.float64(p.astype(np.float64).sum()), np.float64(v.astype(np.float64).sum())
        if s == cfg["epochs"] * (cfg["n"] // cfg["B"]) - 1:
            out[f"step{s}.policy"], out[f"step{s}.value"] = p, v
        step_no[0] += 1
        return r
    learner.value_optimizer.step = rec
    # the batch gradient of the FIRST optimiser step as clip_grad_norm_ receives it (ppo_learner.py:187-190: the value net first,
    # then the policy), before it is scaled: every 8th entry + norms (the object the 1e-5 gradient tolerance is about; parameters
    # after Adam only show it through a scale-free, ill-conditioned step)
    real_clip, seen = torch.nn.utils.clip_grad_norm_, []

    def spy(parameters, max_norm, *a, **k):
        parameters = list(parameters)
        if len(seen) < 2:
            seen.append(torch.cat([p.grad.detach().reshape(-1) for p in parameters]).clone().numpy())
        return real_clip(parameters, max_norm, *a, **k)
    torch.nn.utils.clip_grad_norm_ = spy
    try:
        report = learner.learn(buf)
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
    for tag, gvec in zip(("value", "policy"), seen):
        out[f"grad0.{tag}_every8"] = gvec[::8]
        out[f"grad0.{tag}_l2"] = np.float64(np.sqrt((gvec.astype(np.float64) ** 2).sum()))
        out[f"grad0.{tag}_max"] = np.float64(np.abs(gvec).max())
    report.pop("PPO Batch Consumption Time")
    out.update({"report." + k: np.float64(v) for k, v in report.items()})
    save("g5big_learn_discrete_256x3", **out, n_steps=np.int64(step_no[0]), cfg=np.array(json.dumps(cfg)))


def g5bigb_inputs(cfg, actions, log_probs):
    """The experience of G5big-b [r5]: a WELL-CONDITIONED workload at the paired-launch size.  States, advantages and value targets
    come from seeds alone with IEEE adds / multiplies of numpy legacy draws (the same bits on every host: a test rebuilds them);
    `actions` (what the reference's policy sampled at its initial weights) and `log_probs` (that sample's log-probability + N(0, 0.1),
    nudged off the clip edges) come from the fixture, which holds them (uint8 / float32).  Advantages carry a per-action and a
    per-state signal (c[a] + 0.7 sign(s[a]) + 0.3 noise) and the targets a function of the state, so the batch gradient is a
    coherent sum, not a few outliers over cancelling noise as in G5big.  Returned in submit_experience order."""
    rs = np.random.RandomState(cfg["seed"] + 1)
    n, d, A = cfg["n"], cfg["d"], cfg["n_act"]
    states = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    rewards = rs.randn(n).astype(np.float32)
    dones = (rs.rand(n) < 0.03).astype(np.float32)
    trunc = ((rs.rand(n) < 0.03) & (dones == 0)).astype(np.float32)
    c = 0.5 * rs.randn(A)
    noise_a, noise_v = rs.randn(n), rs.randn(n)
    a = np.asarray(actions).astype(np.int64)
    s_a = np.sign(states[np.arange(n), a].astype(np.float64))
    adv = (c[a] + 0.7 * s_a + 0.3 * noise_a).astype(np.float32)
    values = (1.5 * np.sign(states[:, 0].astype(np.float64)) + 0.5 * states[:, 1].astype(np.float64) + 0.3 * noise_v).astype(np.float32)
    return (states, a.astype(np.float32), np.asarray(log_probs, np.float32), rewards, states[:1].repeat(n, 0), dones, trunc, values, adv)


def g5bigb_learn(R):
    """G5big-b [r5]: the size of G5big (256x3, n = B = 262,144, MB = 65,536, 2 optimiser steps: paired + gather-fused launches)
    on a well-conditioned workload (g5bigb_inputs), so that the product's first-step batch gradient can be held to the REFERENCE's
    own gradient directly -- 1e-5 of max|g|, the north star's number -- instead of through a float64 yardstick.  Actions are sampled
    by the reference's policy at its initial weights (DiscreteFF.get_action, discrete_policy.py:44-62, in 4 chunks), the old
    log-probabilities are that sample's + N(0, 0.1); rows whose first-step ratio would sit within 1e-3 of a clip edge (where one
    float32 rounding decides whether the row's gradient is kept: the G4 fixture's knife-edge rows) get their old log-probability
    moved by 0.01.  Stored: the actions (uint8), the log-probabilities, and the same outputs as G5big."""
    cfg = dict(policy_type=0, d=107, n_act=90, layers=(256, 256, 256), n=262144, B=262144, MB=65536, epochs=2, seed=4321,
               lr=3e-4, clip=0.2, ent=0.005)
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    learner = R["PPOLearner"](cfg["d"], cfg["n_act"], 0, cfg["layers"], cfg["layers"], (0.1, 1.0), cfg["B"], cfg["epochs"], cfg["lr"],
                              cfg["lr"], cfg["clip"], cfg["ent"], cfg["MB"], "cpu")
    vec = lambda m: torch.nn.utils.parameters_to_vector(m.parameters()).detach().clone().numpy()
    out = {"p0.hash": param_hash(vec(learner.policy)), "v0.hash": param_hash(vec(learner.value_net)),
           "p0.head": vec(learner.policy)[:64], "v0.head": vec(learner.value_net)[:64]}
    states = g5bigb_inputs(cfg, np.zeros(cfg["n"], np.uint8), np.zeros(cfg["n"], np.float32))[0]
    acts, lps = [], []
    with torch.no_grad():
        for s in range(0, cfg["n"], 65536):
            a, lp = learner.policy.get_action(states[s:s + 65536])
            acts.append(a.numpy().reshape(-1))
            lps.append(lp.numpy().reshape(-1).astype(np.float64))
    actions, lp_pol = np.concatenate(acts).astype(np.uint8), np.concatenate(lps)
    rs = np.random.RandomState(cfg["seed"] + 2)
    old = lp_pol + 0.1 * rs.randn(cfg["n"])
    ratio = np.exp(lp_pol - old)
    edge = (np.abs(ratio - (1 - cfg["clip"])) < 1e-3) | (np.abs(ratio - (1 + cfg["clip"])) < 1e-3)
    old[edge] += 0.01
    ratio = np.exp(lp_pol - old.astype(np.float32).astype(np.float64))
    assert not ((np.abs(ratio - (1 - cfg["clip"])) < 5e-4) | (np.abs(ratio - (1 + cfg["clip"])) < 5e-4)).any()
    log_probs = old.astype(np.float32)
    out["exp.actions_u8"], out["exp.log_probs"] = actions, log_probs
    out["exp.first_step_clip_fraction"] = np.float64(((ratio < 1 - cfg["clip"]) | (ratio > 1 + cfg["clip"])).mean())
    buf = R["ExperienceBuffer"](cfg["n"], cfg["seed"], "cpu")
    exp = g5bigb_inputs(cfg, actions, log_probs)
    out["exp.hash"] = np.asarray([param_hash(x.reshape(-1)) for x in exp[:3] + exp[7:]])
    buf.submit_experience(*exp)
    step_no = [0]
    orig = learner.value_optimizer.step

    def rec(*a, **k):  # the value optimiser steps last (ppo_learner.py:192-193)
        r = orig(*a, **k)
        s = step_no[0]
        p, v = vec(learner.policy), vec(learner.value_net)
        out[f"step{s}.policy_head"], out[f"step{s}.value_head"] = p[:64], v[:64]
        out[f"step{s}.policy_sum"], out[f"step{s}.value_sum"] = np.float64(p.astype(np.float64).sum()), np.float64(v.astype(np.float64).sum())
        if s == cfg["epochs"] * (cfg["n"] // cfg["B"]) - 1:
            out[f"step{s}.policy"], out[f"step{s}.value"] = p, v
        step_no[0] += 1
        return r
    learner.value_optimizer.step = rec
    real_clip, seen = torch.nn.utils.clip_grad_norm_, []

    def spy(parameters, max_norm, *a, **k):  # the first step's batch gradient as clip_grad_norm_ receives it (value net first)
        parameters = list(parameters)
        if len(seen) < 2:
            seen.append(torch.cat([p.grad.detach().reshape(-1) for p in parameters]).clone().numpy())
        return real_clip(parameters, max_norm, *a, **k)
    torch.nn.utils.clip_grad_norm_ = spy
    try:
        report = learner.learn(buf)
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
    for tag, gvec in zip(("value", "policy"), seen):
        out[f"grad0.{tag}_every8"] = gvec[::8]
        out[f"grad0.{tag}_l2"] = np.float64(np.sqrt((gvec.astype(np.float64) ** 2).sum()))
        out[f"grad0.{tag}_max"] = np.float64(np.abs(gvec).max())
    report.pop("PPO Batch Consumption Time")
    out.update({"report." + k: np.float64(v) for k, v in report.items()})
    save("g5bigb_learn_discrete_256x3", **out, n_steps=np.int64(step_no[0]), cfg=np.array(json.dumps(cfg)))


def make_gae_inputs(n, seed, trunc_dtype):
    rs = np.random.RandomState(seed)
    rews = rs.randn(n).astype(np.float32) * 2.0
    values = rs.randn(n + 1).astype(np.float32)
    dones = np.zeros(n, np.float32)
    trunc = np.zeros(n, trunc_dtype)
    # ragged trajectories; every trajectory end is done or truncated (quirk Q4), plus mid-trajectory dones
    t = 0
    while t < n:
        ln = int(rs.randint(1, 60))
        end = min(t + ln, n) - 1
        if rs.rand() < 0.5:
            dones[end] = 1
        else:
            trunc[end] = 1
        t = end + 1
    mid = rs.rand(n) < 0.02
    dones[mid] = 1
    trunc[dones == 1] = 0
    return rews, dones, trunc, values


def g3_gae(R):
    """G3: compute_gae (torch_functions.py:36-78) with the exact input types learner.py:352-366 passes."""
    out = {}
    case = 0
    for trunc_dtype in (np.float64, np.float32):
        for ret_std in (None, np.float32(1.7), np.float32(1e-3)):
            for n in (1, 7, 512):
                rews, dones, trunc, values = make_gae_inputs(n, 100 + case, trunc_dtype)
                vlist = torch.as_tensor(values).flatten().tolist()  # python floats, as learner.py:352
                vt, adv, rets = R["torch_functions"].compute_gae(rews, dones, trunc, vlist, gamma=0.99, lmbda=0.95,
                                                                return_std=ret_std)
                pre = f"c{case}."
                out[pre + "rews"] = rews
                out[pre + "dones"] = dones
                out[pre + "trunc"] = trunc
                out[pre + "values"] = values
                out[pre + "ret_std"] = np.float32(np.nan) if ret_std is None else ret_std
                out[pre + "value_targets"] = vt
                out[pre + "advantages"] = adv
                out[pre + "returns"] = np.asarray([float(r) for r in rets], np.float64)
                case += 1
    # one more with non-default gamma/lambda = 1 (no decay: exercises the long-range carry)
    rews, dones, trunc, values = make_gae_inputs(300, 999, np.float64)
    dones[:] = 0
    trunc[:] = 0
    trunc[-1] = 1
    vt, adv, rets = R["torch_functions"].compute_gae(rews, dones, trunc, torch.as_tensor(values).tolist(),
                                                    gamma=1.0, lmbda=1.0, return_std=None)
    pre = f"c{case}."
    out.update({pre + "rews": rews, pre + "dones": dones, pre + "trunc": trunc, pre + "values": values,
                pre + "ret_std": np.float32(np.nan), pre + "value_targets": vt, pre + "advantages": adv,
                pre + "returns": np.asarray([float(r) for r in rets], np.float64),
                pre + "gamma": np.float64(1.0), pre + "lmbda": np.float64(1.0)})
    out["n_cases"] = np.int64(case + 1)
    save("g3_gae", **out)


def _loss_and_grads(R, learner_like, obs, acts, old_logp, adv, targets, clip, ent_coef, mb_ratio):
    """The per-minibatch body of PPOLearner.learn (ppo_learner.py:146-185) run with the reference's modules."""
    policy, value_net = learner_like
    policy.zero_grad()
    value_net.zero_grad()
    vals = value_net(obs).view_as(targets)
    log_probs, entropy = policy.get_backprop_data(obs, acts)
    log_probs = log_probs.view_as(old_logp)
    ratio = torch.exp(log_probs - old_logp)
    clipped = torch.clamp(ratio, 1.0 - clip, 1.0 + clip)
    with torch.no_grad():
        log_ratio = log_probs - old_logp
        kl = ((torch.exp(log_ratio) - 1) - log_ratio).mean()
        clip_fraction = torch.mean((torch.abs(ratio - 1) > clip).float())
    policy_loss = -torch.min(ratio * adv, clipped * adv).mean()
    value_loss = torch.nn.MSELoss()(vals, targets) * mb_ratio
    ppo_loss = (policy_loss - entropy * ent_coef) * mb_ratio
    ppo_loss.backward()
    value_loss.backward()
    return dict(logp=log_probs.detach(), entropy=entropy.detach(), ratio=ratio.detach(), kl=kl, clip_fraction=clip_fraction,
                policy_loss=policy_loss.detach(), value_loss=(value_loss / mb_ratio).detach(), vals=vals.detach())


def g4_discrete_loss(R):
    """G4: one minibatch of loss + grads, ratios engineered to hit in-range / clipped / tie, and a prob < 1e-11."""
    torch.manual_seed(21)
    pol = R["DiscreteFF"](107, 90, (32, 32), "cpu")
    val = R["ValueEstimator"](107, (32, 32), "cpu")
    with torch.no_grad():
        pol.model[4].weight.mul_(60.0)  # spread logits so that some clamped probs sit at 1e-11
    rs = np.random.RandomState(7)
    n = 96
    obs = torch.as_tensor(np.clip(rs.randn(n, 107), -5, 5).astype(np.float32))
    with torch.no_grad():
        probs = torch.clamp(pol.get_output(obs), 1e-11, 1)
        acts = torch.multinomial(probs, 1, True)
        # force a few chosen actions to be ones whose prob is clamped (gradient-dead region)
        low = probs.argmin(-1)
        acts[:8, 0] = low[:8]
        logp = torch.log(probs).gather(-1, acts).flatten()
    target_ratio = np.ones(n, np.float32)
    target_ratio[8:24] = 0.5   # clipped low
    target_ratio[24:40] = 1.5  # clipped high
    target_ratio[40:56] = 0.8  # exactly on the lower clip edge (1 - 0.2 in fp32 arithmetic is not exact: near-tie)
    target_ratio[56:72] = 1.2
    # rows 72.. keep ratio == 1 exactly: torch.min tie
    old_logp = (logp - torch.log(torch.as_tensor(target_ratio))).float()
    adv = torch.as_tensor(rs.randn(n).astype(np.float32) * 2)
    targets = torch.as_tensor(rs.randn(n).astype(np.float32))
    res = _loss_and_grads(R, (pol, val), obs, acts.float(), old_logp, adv, targets, 0.2, 0.005, 0.25)
    save("g4_discrete_loss", obs=obs, acts=acts.float(), old_logp=old_logp, adv=adv, targets=targets,
         clip=np.float32(0.2), ent_coef=np.float32(0.005), mb_ratio=np.float32(0.25), probs=probs,
         **{"out." + k: v for k, v in res.items()}, **params_of(pol, "p."), **params_of(val, "v."),
         **grads_of(pol, "gp."), **grads_of(val, "gv."))


def _synthetic_experience(rs, n, d, policy, act_shape):
    states = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    with torch.no_grad():
        a, lp = policy.get_action(states)
    actions = a.numpy().astype(np.float32).reshape((n,) + act_shape)
    log_probs = lp.numpy().astype(np.float32)
    rewards = rs.randn(n).astype(np.float32)
    next_states = np.clip(rs.randn(n, d), -5, 5).astype(np.float32)
    dones = (rs.rand(n) < 0.03).astype(np.float32)
    trunc = ((rs.rand(n) < 0.03) & (dones == 0)).astype(np.float32)
    values = rs.randn(n).astype(np.float32)
    adv = rs.randn(n).astype(np.float32)
    return states, actions, log_probs, rewards, next_states, dones, trunc, values, adv


def _run_learn(R, policy_type, d, n_act, layers, n, B, MB, epochs, seed, tag, act_shape=()):
    """G5/G9: full PPOLearner.learn on a synthetic buffer; records params after each optimiser step."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    learner = R["PPOLearner"](d, n_act, policy_type, layers, layers, (0.1, 1.0), B, epochs, 3e-4, 3e-4, 0.2, 0.005, MB, "cpu")
    init = {**params_of(learner.policy, "p0."), **params_of(learner.value_net, "v0.")}
    buf = R["ExperienceBuffer"](n, seed, "cpu")
    rs = np.random.RandomState(seed + 1)
    exp = _synthetic_experience(rs, n, d, learner.policy, act_shape)
    buf.submit_experience(*exp)
    snaps = {}
    step_no = [0]
    orig = learner.value_optimizer.step

    def rec(*a, **k):  # value optimizer steps last (ppo_learner.py:192-193): snapshot after it
        r = orig(*a, **k)
        s = step_no[0]
        vec_p = torch.nn.utils.parameters_to_vector(learner.policy.parameters()).detach().clone()
        vec_v = torch.nn.utils.parameters_to_vector(learner.value_net.parameters()).detach().clone()
        snaps[f"step{s}.policy"] = vec_p
        snaps[f"step{s}.value"] = vec_v
        step_no[0] += 1
        return r

    learner.value_optimizer.step = rec
    report = learner.learn(buf)
    report.pop("PPO Batch Consumption Time")
    names = ["states", "actions", "log_probs", "rewards", "next_states", "dones", "truncated", "values", "advantages"]
    rep = {"report." + k: np.float64(v) for k, v in report.items()}
    opt_state = learner.policy_optimizer.state_dict()
    save(tag, **init, **{"exp." + k: v for k, v in zip(names, exp)}, **snaps, **rep, n_steps=np.int64(step_no[0]),
         cfg=np.array(json.dumps(dict(policy_type=policy_type, d=d, n_act=n_act, layers=layers, n=n, B=B, MB=MB,
                                      epochs=epochs, seed=seed, lr=3e-4, clip=0.2, ent=0.005))),
         adam_step=np.float64(float(opt_state["state"][0]["step"])),
         adam_exp_avg0=opt_state["state"][0]["exp_avg"], adam_exp_avg_sq0=opt_state["state"][0]["exp_avg_sq"])
    return learner


def g5_learn(R):
    _run_learn(R, 0, 107, 90, (32, 32), 1024, 512, 128, 2, 123, "g5_learn_discrete")


def g6_shuffle(R):
    """G6: ExperienceBuffer.get_all_batches_shuffled index stream (experience_buffer.py:89-102)."""
    buf = R["ExperienceBuffer"](1000, 123, "cpu")
    n = 1000
    ar = np.arange(n, dtype=np.float32)
    buf.submit_experience(ar[:, None].repeat(3, 1), ar, ar, ar, ar[:, None].repeat(3, 1), ar, ar, ar, ar)
    out = {}
    for epoch in range(2):
        got = [b[0].numpy().astype(np.int64) for b in buf.get_all_batches_shuffled(300)]
        out[f"epoch{epoch}"] = np.stack(got)  # 3 batches of 300; the last 100 indices are dropped (quirk Q7)
    buf.clear()  # re-seeds (experience_buffer.py:104-118)
    buf.submit_experience(ar[:, None].repeat(3, 1), ar, ar, ar, ar[:, None].repeat(3, 1), ar, ar, ar, ar)
    out["after_clear"] = np.stack([b[0].numpy().astype(np.int64) for b in buf.get_all_batches_shuffled(300)])
    # raw permutations at the benchmark size, checksummed
    rs = np.random.RandomState(123)
    p = rs.permutation(524288)
    out["perm524288_head"] = p[:64]
    out["perm524288_xor"] = np.bitwise_xor.reduce(p * np.arange(1, p.size + 1))
    out["perm524288_wsum"] = np.sum((p * (np.arange(p.size) % 1000003)) % 2147483647)
    p2 = rs.permutation(524288)
    out["perm524288_second_head"] = p2[:64]
    save("g6_shuffle", **out)


def g7_welford(R):
    """G7: WelfordRunningStat (running_stats.py:15-137) as driven by learner.py:368-372."""
    rs = np.random.RandomState(3)
    st = R["WelfordRunningStat"](1)
    rets = (rs.randn(300) * 3 + 1).astype(np.float64)
    out = {"returns": rets, "std_initial": st.std.copy(), "mean_initial": st.mean.copy()}
    st.increment(list(rets[:1]), 1)
    out["std_after1"] = st.std.copy()
    st.increment(list(rets[1:150]), 149)
    out["mean150"], out["std150"] = st.mean.copy(), st.std.copy()
    st.increment(list(rets[150:]), 150)
    out["mean300"], out["std300"] = st.mean.copy(), st.std.copy()
    js = st.to_json()
    out["json_mean"], out["json_var"], out["json_count"] = np.asarray(js["mean"]), np.asarray(js["var"]), np.int64(js["count"])
    # vector obs stats (shape (5,)) incl. the increment(samples, num) row loop (batched_agent_manager.py:377-380)
    ob = R["WelfordRunningStat"](5)
    obs = rs.randn(12, 5).astype(np.float32)
    ob.increment(obs[:4], 4)
    ob.increment(obs[4:], 8)
    out["obs"], out["obs_mean"], out["obs_std"] = obs, ob.mean.copy(), ob.std.copy()
    # merge path (running_stats.py:71-98)
    a, b = R["WelfordRunningStat"](5), R["WelfordRunningStat"](5)
    a.increment(obs[:5], 5)
    b.increment(obs[5:], 7)
    a.increment_from_serialized_other(b.serialize())
    out["merged_mean"], out["merged_var"], out["merged_count"] = a.running_mean.copy(), a.running_variance.copy(), np.int64(a.count)
    save("g7_welford", **out)


def g8_fifo(R):
    """G8: ExperienceBuffer._cat four cases (experience_buffer.py:18-37) through submit_experience."""
    out = {}

    def run(tag, size, chunks):
        buf = R["ExperienceBuffer"](size, 1, "cpu")
        base = 0
        for c in chunks:
            ar = np.arange(base, base + c, dtype=np.float32)
            base += c
            buf.submit_experience(ar[:, None].repeat(2, 1), ar, ar, ar, ar[:, None].repeat(2, 1), ar, ar, ar, ar)
        out[tag + ".rewards"] = buf.rewards
        out[tag + ".states"] = buf.states
        out[tag + ".chunks"] = np.asarray(chunks)
        out[tag + ".size"] = np.int64(size)

    run("under", 10, [3, 4])           # t1+t2 <= size
    run("exact", 10, [3, 10])          # len(t2) == size
    run("over", 10, [6, 7])            # t1+t2 > size
    run("huge", 10, [4, 25])           # len(t2) > size
    run("stream", 10, [4, 4, 4, 4, 1, 12, 3])
    save("g8_fifo", **out)


def g9_other_heads(R):
    """G9: continuous (d=231, k=8) and multi-discrete versions of G1/G4 (+ a short learn())."""
    # ---- continuous forward/sample
    torch.manual_seed(31)
    pol = R["ContinuousPolicy"](231, 16, (48, 48), "cpu", var_min=0.1, var_max=1.0)
    val = R["ValueEstimator"](231, (48, 48), "cpu")
    rs = np.random.RandomState(9)
    obs = torch.as_tensor(np.clip(rs.randn(80, 231), -5, 5).astype(np.float32))
    with torch.no_grad():
        mean, std = pol.get_output(obs)
        st = torch.get_rng_state()
        act, lp = pol.get_action(obs)
        torch.set_rng_state(st)
        eps = torch.empty(80, 8).normal_(0, 1)  # Normal.sample consumes normal_(mean=0,std=1) then mean+std*eps
        det_mean, _ = pol.get_action(obs, deterministic=True)
    out = dict(obs=obs, mean=mean, std=std, eps=eps, act=act, logp=lp, det=det_mean)
    # engineered old log-probs for the loss
    tr = np.ones(80, np.float32)
    tr[:20] = 0.6
    tr[20:40] = 1.4
    old = (lp - torch.log(torch.as_tensor(tr))).float()
    adv = torch.as_tensor(rs.randn(80).astype(np.float32))
    tg = torch.as_tensor(rs.randn(80).astype(np.float32))
    res = _loss_and_grads(R, (pol, val), obs, act, old, adv, tg, 0.2, 0.005, 0.5)
    save("g9_continuous", **out, old_logp=old, adv=adv, targets=tg, clip=np.float32(0.2), ent_coef=np.float32(0.005),
         mb_ratio=np.float32(0.5), **{"out." + k: v for k, v in res.items()}, **params_of(pol, "p."),
         **params_of(val, "v."), **grads_of(pol, "gp."), **grads_of(val, "gv."))

    # ---- multi-discrete
    torch.manual_seed(41)
    pol = R["MultiDiscreteFF"](107, (32, 32), "cpu")
    val = R["ValueEstimator"](107, (32, 32), "cpu")
    obs = torch.as_tensor(np.clip(rs.randn(72, 107), -5, 5).astype(np.float32))
    with torch.no_grad():
        logits = pol.get_output(obs)
        st = torch.get_rng_state()
        act, lp = pol.get_action(obs)
        torch.set_rng_state(st)
        q = torch.empty(72 * 8, 3).exponential_(1)  # Categorical.sample -> multinomial on [n*8, 3] probs
        det, _ = pol.get_action(obs, deterministic=True)
    tr = np.ones(72, np.float32)
    tr[:18] = 0.6
    tr[18:36] = 1.4
    old = (lp - torch.log(torch.as_tensor(tr))).float()
    adv = torch.as_tensor(rs.randn(72).astype(np.float32))
    tg = torch.as_tensor(rs.randn(72).astype(np.float32))
    res = _loss_and_grads(R, (pol, val), obs, act, old, adv, tg, 0.2, 0.005, 0.5)
    save("g9_multidiscrete", obs=obs, logits=logits, q=q, act=act, logp=lp, det=np.asarray(det), old_logp=old, adv=adv,
         targets=tg, clip=np.float32(0.2), ent_coef=np.float32(0.005), mb_ratio=np.float32(0.5),
         **{"out." + k: v for k, v in res.items()}, **params_of(pol, "p."), **params_of(val, "v."),
         **grads_of(pol, "gp."), **grads_of(val, "gv."))

    _run_learn(R, 2, 231, 8, (48, 48), 512, 256, 128, 2, 77, "g9_learn_continuous", act_shape=(8,))
    _run_learn(R, 1, 107, 8, (32, 32), 512, 256, 128, 2, 78, "g9_learn_multidiscrete", act_shape=(8,))


def g10_checkpoint(R):
    """G10: state_dict key lists + Adam state layout after 2 optimiser steps (ppo_learner.py:240-271)."""
    torch.manual_seed(5)
    learner = R["PPOLearner"](107, 90, 0, (32, 32), (32, 32), (0.1, 1.0), 128, 2, 3e-4, 3e-4, 0.2, 0.005, 64, "cpu")
    buf = R["ExperienceBuffer"](128, 5, "cpu")
    rs = np.random.RandomState(6)
    buf.submit_experience(*_synthetic_experience(rs, 128, 107, learner.policy, ()))
    learner.learn(buf)
    sd = learner.policy_optimizer.state_dict()
    meta = dict(policy_keys=list(learner.policy.state_dict().keys()),
                value_keys=list(learner.value_net.state_dict().keys()),
                adam_param_group_keys=sorted(sd["param_groups"][0].keys()),
                adam_param_group={k: (v if not isinstance(v, (tuple, list)) else list(v))
                                  for k, v in sd["param_groups"][0].items()},
                adam_state_keys=sorted(sd["state"][0].keys()),
                adam_state_ids=sorted(sd["state"].keys()),
                adam_step_dtype=str(sd["state"][0]["step"].dtype),
                adam_step_value=float(sd["state"][0]["step"]),
                cumulative_model_updates=learner.cumulative_model_updates)
    save("g10_checkpoint", meta=np.array(json.dumps(meta)))


def g11_wire(R):
    """G11: what the reference's env worker (batched_agents/batched_agent.py:4-222) puts on the wire -- every datagram and,
    after every step, its slab of the shared array -- for scripted action sequences (oracle/host.py::drive_worker is the
    scripted learner side).  Two cases: a 2-agent environment without a metrics function, and a one-agent environment with
    rank-1 observations, a scalar reward and a metrics function; both cross episode ends (reset) and a truncation."""
    sys.path.insert(0, os.path.dirname(HERE))                   # tests/: synthetic_env
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # the repository root: oracle/
    import synthetic_env
    from oracle import host
    gym = sys.modules["gym"]  # the worker imports gym for its action-space type checks: give the stub the names it reads
    gym.spaces = types.SimpleNamespace(multi_discrete=types.SimpleNamespace(MultiDiscrete=synthetic_env.MultiDiscrete),
                                       box=types.SimpleNamespace(Box=synthetic_env.Box))
    from rlgym_ppo.batched_agents.batched_agent import batched_agent_process
    out = {}
    rs = np.random.RandomState(11)
    cases = (("multi", synthetic_env.make_wire_env, None, [rs.randint(0, 7, (2, 1)).astype(np.float32) for _ in range(14)]),
             ("single", synthetic_env.make_single_env, synthetic_env.step_count_metrics,
              [rs.randint(0, 5, (1, 1)).astype(np.float32) for _ in range(9)]))
    for tag, env_fn, metrics_fn, actions in cases:
        rec = host.drive_worker(batched_agent_process, env_fn, metrics_fn, actions)
        out[tag + ".reset"], out[tag + ".shapes"] = rec["reset"], rec["shapes"]
        out[tag + ".n_steps"] = len(actions)
        out[tag + ".actions"] = np.stack(actions)
        for i, (h, sl) in enumerate(zip(rec["step_headers"], rec["slabs"])):
            out[f"{tag}.hdr{i}"], out[f"{tag}.slab{i}"] = h, sl
    save("g11_wire", **out)


def main():
    only = sys.argv[1:]  # e.g. `make_golden.py g5big_learn`: regenerate the named fixtures only
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: fixtures can only be regenerated in the build container")
    R = _import_reference()
    torch.set_num_threads(1)  # deterministic reduction order in the CPU GEMMs that produce the fixtures
    for fn in (g1_g2_forward, g1bc_forward_fused_shapes, g3_gae, g4_discrete_loss, g5_learn, g5big_learn, g5bigb_learn, g6_shuffle, g7_welford, g8_fifo,
               g9_other_heads, g10_checkpoint, g11_wire):
        if not only or fn.__name__ in only:
            fn(R)


if __name__ == "__main__":
    main()
