"""oracle/ppo.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

torch-CPU fp32 restatement of `PPOLearner.learn` (reference: rlgym_ppo/ppo/ppo_learner.py:92-238) on
plain parameter lists, plus a float64 numpy form with hand-derived gradients (the formulas the HIP
backward kernels implement; DESIGN.md "Loss-epilogue gradients").

    minibatch_autograd(...)   losses/stats/grads of one minibatch through torch autograd -- the same
                              ATen op sequence as ppo_learner.py:146-180, so on CPU it is the reference.
    minibatch_analytic(...)   the same quantities in float64 with explicit backprop.
    clip_coef / adam_step     clip_grad_norm_(max_norm=0.5) and torch.optim.Adam defaults, written out
                              (ppo_learner.py:56-59,187-193).
    learn(...)                the full epoch/batch/minibatch loop incl. the shuffle
                              (experience_buffer.py:89-102) and the 8-key report (ppo_learner.py:225-234).
"""
import math

import numpy as np
import torch

from . import nets

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8
MAX_GRAD_NORM = 0.5


# --------------------------------------------------------------------------------- one minibatch, fp32
def minibatch_autograd(head, pol, val, obs, acts, old_logp, adv, targets, clip, ent_coef, mb_ratio,
                       var_range=(0.1, 1.0)):
    pol = [(w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)) for w, b in pol]
    val = [(w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)) for w, b in val]
    obs = nets.as_obs(obs)
    vals = nets.value_forward(val, obs).view_as(targets)
    logp, entropy = nets.backprop(head, pol, obs, acts, var_range)
    logp = logp.view_as(old_logp)
    ratio = torch.exp(logp - old_logp)
    clipped = torch.clamp(ratio, 1.0 - clip, 1.0 + clip)
    with torch.no_grad():
        log_ratio = logp - old_logp
        kl = ((torch.exp(log_ratio) - 1) - log_ratio).mean().item()
        clip_fraction = torch.mean((torch.abs(ratio - 1) > clip).float()).item()
    policy_loss = -torch.min(ratio * adv, clipped * adv).mean()
    value_loss = torch.nn.functional.mse_loss(vals, targets) * mb_ratio
    ppo_loss = (policy_loss - entropy * ent_coef) * mb_ratio
    ppo_loss.backward()
    value_loss.backward()
    return dict(logp=logp.detach(), vals=vals.detach(), entropy=entropy.item(), kl=kl, clip_fraction=clip_fraction,
                policy_loss=policy_loss.item(), value_loss=(value_loss / mb_ratio).item(),
                grad_policy=[(w.grad, b.grad) for w, b in pol], grad_value=[(w.grad, b.grad) for w, b in val])


# ------------------------------------------------------------------------------ one minibatch, float64
def _fwd64(params, x, out_tanh=False, masks=None, detail=None):
    """float64 forward.  `masks` (optional): one boolean [n, H_l] array per hidden layer that REPLACES the ReLU decision
    [pre > 0] -- the decisions a float32 implementation actually took; where they differ from float64's own, |pre| is within
    float32 rounding of 0, so the forward value changes by no more than that and the backward pass mirrors the
    implementation's choice exactly.  `detail` (optional dict) receives the pre-activations and the float32 rounding scale
    sum_k |x_k w_jk| + |b_j| of every hidden unit (what bounds a legitimate sign flip)."""
    acts = [x]
    used = []
    h = x
    for i, (w, b) in enumerate(params):
        pre = h @ w.T + b
        if i < len(params) - 1:
            m = (pre > 0.0) if masks is None else np.asarray(masks[i], bool)
            if detail is not None:
                detail.setdefault("pre", []).append(pre)
                detail.setdefault("scale", []).append(np.abs(h) @ np.abs(w).T + np.abs(b))
            h = pre * m
            used.append(m)
        else:
            h = pre
        acts.append(h)
    if out_tanh:
        acts[-1] = np.tanh(acts[-1])
    return acts, used


def _bwd64(params, acts, dout, masks):
    grads = [None] * len(params)
    g = dout
    for i in range(len(params) - 1, -1, -1):
        w, _ = params[i]
        grads[i] = (g.T @ acts[i], g.sum(0))
        if i > 0:
            g = (g @ w) * masks[i - 1]
    return grads


def surrogate_weight(ratio, adv, clip):
    """d min(r*A, clamp(r)*A) / d r, divided by A: torch.min splits a tie 1/2+1/2 and clamp passes the
    gradient on its closed interval."""
    s1 = ratio * adv
    s2 = np.clip(ratio, 1.0 - clip, 1.0 + clip) * adv
    inr = ((ratio >= 1.0 - clip) & (ratio <= 1.0 + clip)).astype(np.float64)
    return np.where(s1 < s2, 1.0, np.where(s1 > s2, inr, 0.5 + 0.5 * inr))


def minibatch_analytic(head, pol, val, obs, acts, old_logp, adv, targets, clip, ent_coef, mb_ratio,
                       var_range=(0.1, 1.0), masks_pol=None, masks_val=None, w_override=None, detail=None):
    """float64 losses / statistics / gradients with hand-derived backprop.  Optional: `masks_pol` / `masks_val` impose the
    ReLU decisions of a float32 implementation (see _fwd64); `w_override` ([n], NaN = keep) imposes its surrogate-branch
    decision on rows whose ratio sits within rounding of a clip edge; `detail` (dict) receives per-row diagnostics."""
    f = lambda t: np.asarray(t, np.float64)
    pol = [(f(w), f(b)) for w, b in pol]
    val = [(f(w), f(b)) for w, b in val]
    x, old, A, tg = f(obs), f(old_logp), f(adv), f(targets)
    # float32-rounded clip edges: the reference compares fp32 ratios against 1.0 -/+ clip evaluated in
    # python floats and cast to fp32 by torch.clamp
    n = x.shape[0]
    dval = None if detail is None else detail.setdefault("val", {})
    dpol = None if detail is None else detail.setdefault("pol", {})
    va, vm = _fwd64(val, x, masks=masks_val, detail=dval)
    v = va[-1][:, 0]
    dv = (mb_ratio * 2.0 * (v - tg) / n)[:, None]
    gv = _bwd64(val, va, dv, vm)

    def weight(ratio):
        w = surrogate_weight(ratio, A, clip)
        if detail is not None:
            detail["ratio"], detail["w"] = ratio, w
        if w_override is not None:
            ov = np.asarray(w_override, np.float64)
            w = np.where(np.isnan(ov), w, ov)
        return w

    value_loss = np.mean((v - tg) ** 2)

    if head == "discrete":
        pa, pm = _fwd64(pol, x, masks=masks_pol, detail=dpol)
        z = pa[-1]
        z = z - z.max(-1, keepdims=True)
        p = np.exp(z)
        p /= p.sum(-1, keepdims=True)
        pc = np.clip(p, nets.PROB_MIN, 1.0)
        lp = np.log(pc)
        a = f(acts).astype(np.int64).reshape(-1)
        rows = np.arange(n)
        logp = lp[rows, a]
        ent_i = -(lp * pc).sum(-1)
        ratio = np.exp(logp - old)
        w = weight(ratio)
        g_pc = mb_ratio * (ent_coef / n) * (lp + 1.0)
        g_pc[rows, a] += mb_ratio * (-(A * w * ratio) / n) / pc[rows, a]
        g_p = g_pc * (p >= nets.PROB_MIN)
        dz = p * (g_p - (g_p * p).sum(-1, keepdims=True))
        gp = _bwd64(pol, pa, dz, pm)
        entropy = ent_i.mean()
    elif head == "gaussian":
        pa, pm = _fwd64(pol, x, out_tanh=True, masks=masks_pol, detail=dpol)
        y = pa[-1]
        k = y.shape[1] // 2
        m, b = nets.var_map(*var_range)
        mu, sd = y[:, :k], y[:, k:] * m + b
        xa = f(acts)
        logp = (-(mu * mu) / (2 * sd * sd) + mu * xa / (sd * sd) - xa * xa / (2 * sd * sd)
                + np.log(1.0 / np.sqrt(2 * np.pi * sd * sd))).sum(-1)
        ratio = np.exp(logp - old)
        w = weight(ratio)
        if detail is not None:  # conditioning of logp on the head's outputs: |d logp / d y_j| (tanh outputs: mean | sd pre-map)
            detail["dlogp_dy"] = np.concatenate([np.abs((xa - mu) / (sd * sd)), np.abs(((xa - mu) ** 2 / sd ** 3 - 1.0 / sd) * m)], 1)
            detail["logp_terms"] = (np.abs(mu * mu / (2 * sd * sd)) + np.abs(mu * xa / (sd * sd)) + np.abs(xa * xa / (2 * sd * sd))
                                    + np.abs(np.log(1.0 / np.sqrt(2 * np.pi * sd * sd)))).sum(-1)
            detail["y"] = y
        g_logp = (mb_ratio * (-(A * w * ratio) / n))[:, None]
        d_mu = g_logp * (xa - mu) / (sd * sd)
        d_sd = g_logp * ((xa - mu) ** 2 / sd ** 3 - 1.0 / sd) - mb_ratio * ent_coef / (n * k) / sd
        dy = np.concatenate([d_mu, d_sd * m], axis=1)
        dz = dy * (1.0 - y * y)
        gp = _bwd64(pol, pa, dz, pm)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + np.log(sd)).mean()
    else:
        pa, pm = _fwd64(pol, x, masks=masks_pol, detail=dpol)
        z = pa[-1]
        a = f(acts).astype(np.int64)
        dz = np.zeros_like(z)
        logp = np.zeros(n)
        ent_i = np.zeros(n)
        s = 0
        parts = []
        for h, bins in enumerate(nets.MD_BINS):
            zz = z[:, s:s + bins]
            zz = zz - zz.max(-1, keepdims=True)
            ls = zz - np.log(np.exp(zz).sum(-1, keepdims=True))
            ph = np.exp(ls)
            logp += ls[np.arange(n), a[:, h]]
            eh = -(ph * ls).sum(-1)
            ent_i += eh
            parts.append((s, bins, ls, ph, eh))
            s += bins
        ratio = np.exp(logp - old)
        w = weight(ratio)
        g_logp = mb_ratio * (-(A * w * ratio) / n)
        for h, (s, bins, ls, ph, eh) in enumerate(parts):
            onehot = np.zeros((n, bins))
            onehot[np.arange(n), a[:, h]] = 1.0
            d = g_logp[:, None] * (onehot - ph)
            d += (-mb_ratio * ent_coef / n) * (-ph * (ls + eh[:, None]))
            dz[:, s:s + bins] = d
        gp = _bwd64(pol, pa, dz, pm)
        entropy = ent_i.mean()

    s1 = ratio * A
    s2 = np.clip(ratio, 1 - clip, 1 + clip) * A
    lr = logp - old
    return dict(logp=logp, vals=v, entropy=entropy, kl=np.mean((np.exp(lr) - 1) - lr),
                clip_fraction=np.mean(np.abs(ratio - 1) > clip), policy_loss=-np.mean(np.minimum(s1, s2)),
                value_loss=value_loss, grad_policy=gp, grad_value=gv)


# -------------------------------------------------------------------------------------- clip and Adam
def clip_coef(grads, max_norm=MAX_GRAD_NORM):
    """torch.nn.utils.clip_grad_norm_: per-tensor 2-norms, 2-norm of those, coef = max/(total+1e-6) clamped to 1."""
    norms = torch.stack([torch.linalg.vector_norm(g, 2) for wb in grads for g in wb])
    total = torch.linalg.vector_norm(norms, 2)
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total


class AdamState:
    def __init__(self, params):
        self.m = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in params]
        self.v = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in params]
        self.step = 0


def adam_step(params, grads, st, lr):
    """torch.optim.Adam(lr) defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad."""
    st.step += 1
    bc1 = 1 - ADAM_B1 ** st.step
    bc2 = 1 - ADAM_B2 ** st.step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for (w, b), (gw, gb), (mw, mb), (vw, vb) in zip(params, grads, st.m, st.v):
        for p, g, m, v in ((w, gw, mw, vw), (b, gb, mb, vb)):
            m.lerp_(g, 1 - ADAM_B1)
            v.mul_(ADAM_B2).addcmul_(g, g, value=1 - ADAM_B2)
            denom = (v.sqrt() / bc2_sqrt).add_(ADAM_EPS)
            p.addcdiv_(m, denom, value=-step_size)


# ------------------------------------------------------------------------------------------ full learn
def learn(head, pol, val, buf, batch_size, mini_batch_size, n_epochs, clip, ent_coef, policy_lr, critic_lr,
          rng, adam_pol=None, adam_val=None, var_range=(0.1, 1.0), on_step=None, rank=0, world=1, allreduce=None):
    """`pol`/`val` are updated in place.  `buf` = dict(states, actions, log_probs, values, advantages) of CPU
    tensors; `rng` a numpy RandomState (persistent across calls, experience_buffer.py:52).  With world > 1 the
    minibatch slices of a batch are dealt to ranks (contiguous blocks when they divide evenly, round-robin otherwise:
    the sum is the same either way) and `allreduce(flat_tensor)` sums across ranks
    (SURVEY.md section 8(e)); world == 1 is the reference loop verbatim."""
    assert batch_size % mini_batch_size == 0
    adam_pol = adam_pol or AdamState(pol)
    adam_val = adam_val or AdamState(val)
    before_p, before_v = nets.flatten(pol).clone(), nets.flatten(val).clone()
    n_iter = n_mb = 0
    s_ent = s_kl = s_vl = 0.0
    clip_fracs = []
    total = buf["advantages"].shape[0]
    mb_ratio = mini_batch_size / batch_size
    for _ in range(n_epochs):
        idx = rng.permutation(total)
        start = 0
        while start + batch_size <= total:
            bi = torch.as_tensor(idx[start:start + batch_size])
            start += batch_size
            b_acts = buf["actions"][bi].view(batch_size, -1)
            b_old, b_obs = buf["log_probs"][bi], buf["states"][bi]
            b_tgt, b_adv = buf["values"][bi], buf["advantages"][bi]
            gp = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in pol]
            gv = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in val]
            stats = torch.zeros(4, dtype=torch.float64)
            for j, s in enumerate(range(0, batch_size, mini_batch_size)):
                n_sl = batch_size // mini_batch_size
                mine = (j // (n_sl // world) == rank) if n_sl % world == 0 else (j % world == rank)
                if not mine:
                    continue
                e = s + mini_batch_size
                acts = b_acts[s:e]
                if head == "discrete":
                    acts = acts.view(-1)
                r = minibatch_autograd(head, pol, val, b_obs[s:e], acts, b_old[s:e], b_adv[s:e], b_tgt[s:e],
                                       clip, ent_coef, mb_ratio, var_range)
                for (aw, ab), (w, b) in zip(gp, r["grad_policy"]):
                    aw += w
                    ab += b
                for (aw, ab), (w, b) in zip(gv, r["grad_value"]):
                    aw += w
                    ab += b
                stats += torch.tensor([r["entropy"], r["kl"], r["value_loss"], r["clip_fraction"]], dtype=torch.float64)
                if world == 1:
                    clip_fracs.append(r["clip_fraction"])
            if world > 1:
                flat = torch.cat([nets.flatten(gp), nets.flatten(gv), stats.float()])
                allreduce(flat)
                np_, nv_ = nets.flatten(gp).numel(), nets.flatten(gv).numel()
                gp = [(w.clone(), b.clone()) for w, b in nets.unflatten(flat[:np_], gp)]
                gv = [(w.clone(), b.clone()) for w, b in nets.unflatten(flat[np_:np_ + nv_], gv)]
                stats = flat[np_ + nv_:].double()
                clip_fracs.append(stats[3].item() / (batch_size // mini_batch_size))
            s_ent += stats[0].item()
            s_kl += stats[1].item()
            s_vl += stats[2].item()
            n_mb += batch_size // mini_batch_size
            cv, _ = clip_coef(gv)
            cp, _ = clip_coef(gp)
            gv = [(w * cv, b * cv) for w, b in gv]
            gp = [(w * cp, b * cp) for w, b in gp]
            adam_step(pol, gp, adam_pol, policy_lr)
            adam_step(val, gv, adam_val, critic_lr)
            n_iter += 1
            if on_step is not None:
                on_step(n_iter - 1, pol, val)
    n_iter_r = max(n_iter, 1)
    n_mb = max(n_mb, 1)
    report = {
        "Cumulative Model Updates": n_iter_r,
        "Policy Entropy": s_ent / n_mb,
        "Mean KL Divergence": s_kl / n_mb,
        "Value Function Loss": s_vl / n_mb,
        "SB3 Clip Fraction": float(np.mean(clip_fracs)) if clip_fracs else 0,
        "Policy Update Magnitude": (before_p - nets.flatten(pol)).norm().item(),
        "Value Function Update Magnitude": (before_v - nets.flatten(val)).norm().item(),
    }
    return report, adam_pol, adam_val


# --------------------------------------------------------------------------------------- full learn, float64
def learn64(head, pol, val, buf, batch_size, mini_batch_size, n_epochs, clip, ent_coef, policy_lr, critic_lr, rng,
            var_range=(0.1, 1.0), on_step=None, weakest=None, on_grad=None):
    """The same loop as learn() evaluated in float64 end to end (minibatch_analytic gradients, clip coefficient, Adam):
    the truth both float32 implementations -- the reference's CPU path and the HIP kernels -- are measured against in
    tests/test_gpu_learner.py.  Returns float64 parameter lists; `on_step(i, pol, val)` after every optimiser step.
    `weakest` (optional dict): receives, per network ("pol" / "val"), a flat array with the smallest |g_i| / max|g| every
    parameter has seen in any optimiser step so far.  Adam's step lr * m / (sqrt(v) + 1e-8) is ill-conditioned in g where
    that ratio is tiny: an absolute gradient error d moves the step by ~lr * d / |g_i|, whatever implementation made it."""
    f = lambda t: np.asarray(t, np.float64).copy()
    pol = [(f(w), f(b)) for w, b in pol]
    val = [(f(w), f(b)) for w, b in val]
    st = {id(p): [np.zeros_like(p), np.zeros_like(p)] for wb in pol + val for p in wb}
    step = 0
    total = len(buf["advantages"])
    mb_ratio = mini_batch_size / batch_size
    B = {k: np.asarray(v, np.float64) for k, v in buf.items()}
    for _ in range(n_epochs):
        idx = rng.permutation(total)
        start = 0
        while start + batch_size <= total:
            bi = idx[start:start + batch_size]
            start += batch_size
            gp = [(np.zeros_like(w), np.zeros_like(b)) for w, b in pol]
            gv = [(np.zeros_like(w), np.zeros_like(b)) for w, b in val]
            for s in range(0, batch_size, mini_batch_size):
                mi = bi[s:s + mini_batch_size]
                acts = B["actions"][mi].reshape(len(mi), -1)
                r = minibatch_analytic(head, pol, val, B["states"][mi], acts[:, 0] if head == "discrete" else acts, B["log_probs"][mi],
                                       B["advantages"][mi], B["values"][mi], clip, ent_coef, mb_ratio, var_range)
                for acc, new in ((gp, r["grad_policy"]), (gv, r["grad_value"])):
                    for (aw, ab), (w, b) in zip(acc, new):
                        aw += w
                        ab += b
            step += 1
            if on_grad is not None:  # the batch gradient clip_grad_norm_ / Adam are about to see (float64), flat per network
                on_grad(step - 1, np.concatenate([g.ravel() for wb in gp for g in wb]), np.concatenate([g.ravel() for wb in gv for g in wb]))
            bc1, bc2 = 1 - ADAM_B1 ** step, 1 - ADAM_B2 ** step
            for key, params, grads, lr in (("val", val, gv, critic_lr), ("pol", pol, gp, policy_lr)):
                if weakest is not None:
                    flat = np.abs(np.concatenate([g.ravel() for wb in grads for g in wb]))
                    rel = flat / max(flat.max(), 1e-300)
                    weakest[key] = rel if key not in weakest else np.minimum(weakest[key], rel)
                total_norm = math.sqrt(sum(float((g * g).sum()) for wb in grads for g in wb))
                coef = min(MAX_GRAD_NORM / (total_norm + 1e-6), 1.0)
                for wb, gwb in zip(params, grads):
                    for p_, g in zip(wb, gwb):
                        g = g * coef
                        m, v = st[id(p_)]
                        m += (1 - ADAM_B1) * (g - m)
                        v *= ADAM_B2
                        v += (1 - ADAM_B2) * g * g
                        p_ -= (lr / bc1) * m / (np.sqrt(v) / math.sqrt(bc2) + ADAM_EPS)
            if on_step is not None:
                on_step(step - 1, pol, val)
    return pol, val
