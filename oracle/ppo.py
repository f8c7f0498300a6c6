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
def _fwd64(params, x, out_tanh=False):
    acts = [x]
    h = x
    for i, (w, b) in enumerate(params):
        h = h @ w.T + b
        if i < len(params) - 1:
            h = np.maximum(h, 0.0)
        acts.append(h)
    if out_tanh:
        acts[-1] = np.tanh(acts[-1])
    return acts


def _bwd64(params, acts, dout):
    grads = [None] * len(params)
    g = dout
    for i in range(len(params) - 1, -1, -1):
        w, _ = params[i]
        grads[i] = (g.T @ acts[i], g.sum(0))
        if i > 0:
            g = (g @ w) * (acts[i] > 0)
    return grads


def surrogate_weight(ratio, adv, clip):
    """d min(r*A, clamp(r)*A) / d r, divided by A: torch.min splits a tie 1/2+1/2 and clamp passes the
    gradient on its closed interval."""
    s1 = ratio * adv
    s2 = np.clip(ratio, 1.0 - clip, 1.0 + clip) * adv
    inr = ((ratio >= 1.0 - clip) & (ratio <= 1.0 + clip)).astype(np.float64)
    return np.where(s1 < s2, 1.0, np.where(s1 > s2, inr, 0.5 + 0.5 * inr))


def minibatch_analytic(head, pol, val, obs, acts, old_logp, adv, targets, clip, ent_coef, mb_ratio,
                       var_range=(0.1, 1.0)):
    f = lambda t: np.asarray(t, np.float64)
    pol = [(f(w), f(b)) for w, b in pol]
    val = [(f(w), f(b)) for w, b in val]
    x, old, A, tg = f(obs), f(old_logp), f(adv), f(targets)
    # float32-rounded clip edges: the reference compares fp32 ratios against 1.0 -/+ clip evaluated in
    # python floats and cast to fp32 by torch.clamp
    n = x.shape[0]
    va = _fwd64(val, x)
    v = va[-1][:, 0]
    dv = (mb_ratio * 2.0 * (v - tg) / n)[:, None]
    gv = _bwd64(val, va, dv)
    value_loss = np.mean((v - tg) ** 2)

    if head == "discrete":
        pa = _fwd64(pol, x)
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
        w = surrogate_weight(ratio, A, clip)
        g_pc = mb_ratio * (ent_coef / n) * (lp + 1.0)
        g_pc[rows, a] += mb_ratio * (-(A * w * ratio) / n) / pc[rows, a]
        g_p = g_pc * (p >= nets.PROB_MIN)
        dz = p * (g_p - (g_p * p).sum(-1, keepdims=True))
        gp = _bwd64(pol, pa, dz)
        entropy = ent_i.mean()
    elif head == "gaussian":
        pa = _fwd64(pol, x, out_tanh=True)
        y = pa[-1]
        k = y.shape[1] // 2
        m, b = nets.var_map(*var_range)
        mu, sd = y[:, :k], y[:, k:] * m + b
        xa = f(acts)
        logp = (-(mu * mu) / (2 * sd * sd) + mu * xa / (sd * sd) - xa * xa / (2 * sd * sd)
                + np.log(1.0 / np.sqrt(2 * np.pi * sd * sd))).sum(-1)
        ratio = np.exp(logp - old)
        w = surrogate_weight(ratio, A, clip)
        g_logp = (mb_ratio * (-(A * w * ratio) / n))[:, None]
        d_mu = g_logp * (xa - mu) / (sd * sd)
        d_sd = g_logp * ((xa - mu) ** 2 / sd ** 3 - 1.0 / sd) - mb_ratio * ent_coef / (n * k) / sd
        dy = np.concatenate([d_mu, d_sd * m], axis=1)
        dz = dy * (1.0 - y * y)
        gp = _bwd64(pol, pa, dz)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + np.log(sd)).mean()
    else:
        pa = _fwd64(pol, x)
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
        w = surrogate_weight(ratio, A, clip)
        g_logp = mb_ratio * (-(A * w * ratio) / n)
        for h, (s, bins, ls, ph, eh) in enumerate(parts):
            onehot = np.zeros((n, bins))
            onehot[np.arange(n), a[:, h]] = 1.0
            d = g_logp[:, None] * (onehot - ph)
            d += (-mb_ratio * ent_coef / n) * (-ph * (ls + eh[:, None]))
            dz[:, s:s + bins] = d
        gp = _bwd64(pol, pa, dz)
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
