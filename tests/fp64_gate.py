"""The parity gate of the PPO minibatch against FLOAT64 TRUTH, with no row excluded (north_star: "fp32 losses/grads within
1e-5 relative").

Two float32 implementations of the same minibatch (the reference's CPU ATen path and the HIP kernels) legitimately differ in
a handful of DISCRETE decisions that the last bit of a float32 sum decides:
  * the ReLU mask of a hidden unit whose pre-activation is within float32 rounding of 0 (one flipped mask moves one row of a
    first-layer weight gradient by O(1/sqrt(mb)) -- far above 1e-5, for ANY pair of float32 implementations);
  * the surrogate branch of a row whose ratio sits within rounding of a clip edge 1 -/+ eps (reference fixture G4 engineers 32).
Instead of deleting such rows, the gate (a) reads the decisions each implementation actually took -- the HIP masks through the
C ABI (the same gemm_nt kernels, layer by layer), the CPU masks from the float32 torch forward, the edge rows' branch from the
output-layer gradient, which is linear in them -- (b) checks that every decision that differs from float64's own is a
legitimately ambiguous one (|pre| below the float32 rounding bound of its dot product; ratio within rounding of the edge), and
(c) compares ALL gradients and statistics with oracle/ppo.py::minibatch_analytic (float64) evaluated under those decisions.
What remains is pure arithmetic error; it is reported next to the same number for the CPU float32 oracle, and the gate is
    err(HIP, fp64) <= max(1e-5, 1.5 * err(CPU fp32 oracle, fp64)).
"""
import ctypes

import numpy as np
import torch

from oracle import nets, ppo

U32 = 2.0 ** -24  # float32 unit roundoff


def _rel(a, b):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def grads_err(got, want):
    """max over parameter tensors of max|got - want| / max|want| (the convention of every gradient check in tests/)."""
    return max(max(_rel(gw, ww), _rel(gb, wb)) for (gw, gb), (ww, wb) in zip(got, want))


def hip_masks(L, params, obs, x3=False):
    """[h_l > 0] of every hidden layer as the HIP forward computes it: the product's own gemm_nt kernels, layer by layer,
    through the C ABI (rlppo_dbg_gemm_nt with the bias+ReLU epilogue on the packed weights).  x3: the split-bf16 update precision --
    layers >= 1 whose shape its kernel covers (width % 256 == 0, contraction % 32 == 0) go through rlppo_dbg_gemm_nt_x3, as in
    rlppo_ppo_minibatch."""
    from rlgym_ppo_amd import _native as N
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dims = [params[0][0].shape[1]] + [w.shape[0] for w, _ in params]
    dims_c = N.dims_array(dims)
    flat = nets.flatten(params).cuda()
    packed = torch.zeros(int(L.rlppo_packed_floats(dims_c, len(params))), device="cuda")
    N.check(L.rlppo_net_pack(st, dims_c, len(params), P(flat), P(packed)))
    x = torch.as_tensor(np.asarray(obs, np.float32)).cuda()
    n, d = x.shape
    pin = int(L.rlppo_padded_width(d))
    h = torch.zeros(n, pin, device="cuda")
    N.check(L.rlppo_pad_rows(st, P(x), 0, n, d, d, P(h), pin, 0, 0.0, 1.0))
    masks, off = [], 0
    for l in range(len(params) - 1):
        pout = int(L.rlppo_padded_out(dims[l + 1]))
        w_ptr = ctypes.c_void_p(packed.data_ptr() + 4 * off)
        b_ptr = ctypes.c_void_p(packed.data_ptr() + 4 * (off + 2 * pout * pin))
        out = torch.empty(n, pout, device="cuda")
        if x3 and l >= 1 and pout % 256 == 0 and pin % 32 == 0:
            planes = torch.zeros(3 * pout * pin, dtype=torch.bfloat16, device="cuda")
            bits = torch.zeros(max(int(L.rlppo_dbg_gemm_nt_bits_bytes(n, pout)), 8), dtype=torch.uint8, device="cuda")
            N.check(L.rlppo_dbg_pack_x3(st, w_ptr, pin, pout, pin, P(planes)))
            N.check(L.rlppo_dbg_gemm_nt_x3(st, P(h), pin, P(planes), b_ptr, P(out), pout, n, pout, pin, 0, P(bits)))
        else:
            N.check(L.rlppo_dbg_gemm_nt(st, P(h), pin, w_ptr, pin, b_ptr, None, 0, P(out), pout, n, pout, pin, 1))
        masks.append((out[:, :dims[l + 1]] > 0).cpu().numpy())
        off += 2 * pout * pin + pout
        h, pin = out, pout
    return masks


def cpu_masks(params, obs):
    h = nets.as_obs(obs)
    masks = []
    for w, b in params[:-1]:
        h = torch.relu(torch.nn.functional.linear(h, w, b))
        masks.append((h > 0).numpy())
    return masks


def _check_flips(params, masks, det, who):
    """Every ReLU decision that differs from float64's own must be one float32 rounding can legitimately flip: |pre| within
    the worst-case rounding bound of its float32 dot product, B_l = (K_l + 2) u (sum_k |h_k w_jk| + |b_j|) + |W_l| B_(l-1)
    (the second term: the layer's inputs carry the previous layer's rounding), with a factor 2 of slack."""
    n_flips, prev = 0, None
    for l, (m, pre, scale) in enumerate(zip(masks, det["pre"], det["scale"])):
        w = np.abs(np.asarray(params[l][0], np.float64))
        bound = (w.shape[1] + 2) * U32 * scale
        if prev is not None:
            bound = bound + prev @ w.T
        prev = bound
        flip = m != (pre > 0.0)
        n_flips += int(flip.sum())
        if flip.any():
            assert (np.abs(pre[flip]) <= 2.0 * bound[flip]).all(), (who, l, float((np.abs(pre[flip]) / bound[flip]).max()))
    return n_flips


def _edge_contributions(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, masks_pol, masks_val, edge, n):
    """Full-branch contribution c_i of every clip-edge row i to the policy gradient (a list over the edge rows of per-layer
    [dW, db] pairs, float64, under the imposed ReLU masks).  The gradient is linear in the branch weights and row i's term
    mb_ratio * (-(A_i w_i ratio_i) / n) reaches the parameters through row i alone, so c_i is evaluated on the SUB-BATCH of the edge
    rows (n' rows with mb_ratio' = mb_ratio n' / n: the same 1 / n scaling) as (w_i = 1) - (w_i = 0) -- what round 5 computed with
    one full-batch float64 pass per edge row (145 s of the suite at 262,144 rows)."""
    sub = lambda a: np.asarray(a)[edge]
    mp = None if masks_pol is None else [np.asarray(m)[edge] for m in masks_pol]
    mv = None if masks_val is None else [np.asarray(m)[edge] for m in masks_val]
    ne = len(edge)
    args = (head, pol, val, sub(obs), sub(acts), sub(old), sub(adv), sub(tgt), clip, ent, mb_ratio * ne / n, var, mp, mv)
    base = ppo.minibatch_analytic(*args, np.zeros(ne))["grad_policy"]
    out = []
    for k in range(ne):
        wk = np.zeros(ne)
        wk[k] = 1.0
        one = ppo.minibatch_analytic(*args, wk)["grad_policy"]
        out.append([(np.asarray(ow, np.float64) - np.asarray(bw, np.float64), np.asarray(ob, np.float64) - np.asarray(bb, np.float64))
                    for (ow, ob), (bw, bb) in zip(one, base)])
    return out


def _edge_decisions(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, masks_pol, masks_val, got_pol, edge, n):
    """Surrogate branch of the rows within rounding of a clip edge, read off the implementation's output-layer gradient (which is
    linear in them): least squares over the edge rows' contributions, rounded to {0, 1}."""
    w0 = np.full(n, np.nan)
    w0[edge] = 0.0
    base = ppo.minibatch_analytic(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, masks_pol, masks_val, w0)
    contrib = _edge_contributions(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, masks_pol, masks_val, edge, n)
    cols = [np.concatenate([c[-1][0].ravel(), c[-1][1]]) for c in contrib]
    A = np.stack(cols, 1)
    gw, gb = got_pol[-1]
    rhs = np.concatenate([np.asarray(gw, np.float64).ravel() - base["grad_policy"][-1][0].ravel(),
                          np.asarray(gb, np.float64) - base["grad_policy"][-1][1]])
    live = np.abs(A).max(0) > 0  # a row with zero advantage contributes nothing either way
    s = np.zeros(len(edge))
    if live.any():
        s[live] = np.linalg.lstsq(A[:, live], rhs, rcond=None)[0]
    # 0 = the clipped branch is the strict minimum, 1 = full gradient, 1/2 = the two float32 PRODUCTS ratio*A and clamp(ratio)*A
    # round to the same number although ratio is just outside the interval: torch.min splits the tie and only the unclipped
    # half carries gradient
    dec = np.round(2.0 * s) / 2.0
    assert (np.abs(s - dec) < 0.02).all() and ((dec >= 0) & (dec <= 1)).all(), ("edge rows are not cleanly decided", s)
    w = np.full(n, np.nan)
    w[edge] = dec
    return w


def gate(L, head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, got, var=(0.1, 1.0), label="", floor=1e-5, x3=False):
    """got = (grad_policy, grad_value, stats[>=5]) of the HIP minibatch over exactly these rows.  Returns the measured errors."""
    gp, gv, stats = got
    obs = np.asarray(obs, np.float32)
    n = obs.shape[0]
    acts_t = torch.as_tensor(np.asarray(acts, np.float32))
    if head == "discrete":
        acts_t = acts_t.view(-1)
    det = {}
    truth = ppo.minibatch_analytic(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, detail=det)
    ratio = det["ratio"]
    lo, hi = float(np.float32(1.0 - clip)), float(np.float32(1.0 + clip))
    # a float32 log-probability of magnitude ~5 carries ~5e-7 of rounding, so a ratio within 1e-5 of an edge may be decided either
    # way by a correct float32 implementation (G4: 32 engineered rows within 3e-6; random data: ~2e-5 * n rows)
    edge = np.flatnonzero(np.minimum(np.abs(ratio - lo), np.abs(ratio - hi)) <= 1e-5)
    cpu = ppo.minibatch_autograd(head, pol, val, torch.as_tensor(obs), acts_t, torch.as_tensor(np.asarray(old, np.float32)),
                                 torch.as_tensor(np.asarray(adv, np.float32)), torch.as_tensor(np.asarray(tgt, np.float32)), clip, ent,
                                 mb_ratio, var)
    out = {"n": n, "edge_rows": len(edge)}
    for who, grads_p, grads_v, mp, mv in (("hip", gp, gv, hip_masks(L, pol, obs, x3), hip_masks(L, val, obs, x3)),
                                          ("cpu", cpu["grad_policy"], cpu["grad_value"], cpu_masks(pol, obs), cpu_masks(val, obs))):
        flips = _check_flips(pol, mp, det["pol"], who) + _check_flips(val, mv, det["val"], who)
        w = None
        if len(edge):
            w = _edge_decisions(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, mp, mv, grads_p, edge, n)
        ref = ppo.minibatch_analytic(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, mp, mv, w)
        out[who] = dict(err=grads_err(list(grads_p) + list(grads_v), ref["grad_policy"] + ref["grad_value"]), flips=flips,
                        ambiguity=grads_err(ref["grad_policy"] + ref["grad_value"], truth["grad_policy"] + truth["grad_value"]),
                        ref=ref, w=w, masks=(mp, mv), grads=(grads_p, grads_v))
    out["edge"] = edge
    e_hip, e_cpu = out["hip"]["err"], out["cpu"]["err"]
    print(f"[fp64 gate] {label or head}: n={n}  err(HIP, fp64)={e_hip:.2e}  err(CPU fp32 oracle, fp64)={e_cpu:.2e}  | ReLU decisions "
          f"differing from fp64: HIP {out['hip']['flips']}, CPU {out['cpu']['flips']} (worth {out['hip']['ambiguity']:.1e} / "
          f"{out['cpu']['ambiguity']:.1e} of the gradient), clip-edge rows {len(edge)}")
    assert e_hip <= max(floor, 1.5 * e_cpu), (label, e_hip, e_cpu)
    # statistics: means over the rows -> float64 truth at 1e-5 (clip fraction: the edge rows may fall either way)
    ref = out["hip"]["ref"]
    if stats is None:  # (a caller that has the gradient only: the first optimiser step of a learn())
        return out
    for name, k, tol in (("entropy", 0, 1e-5), ("kl", 1, 1e-5), ("value_loss", 2, 1e-5), ("policy_loss", 4, 1e-5)):
        want = float(ref[name])
        assert abs(stats[k] - want) <= tol * max(abs(want), 1e-3) + 1e-7, (label, name, stats[k], want)
    assert abs(stats[3] - float(ref["clip_fraction"])) <= (len(edge) + 0.5) / n + 1e-9
    return out


# ----------------------------------------------------------------------------------------------- Gaussian log-probabilities
def gauss_logp_check(params, obs, eps, y_hip, act_hip, logp_hip, var=(0.1, 1.0), label=""):
    """ContinuousPolicy.get_action's log-probability (continuous_policy.py:54-63,87-96) against float64, with a DERIVED bound
    instead of a chosen atol.  The reference's four-term formula  -mu^2/2s^2 + mu x/s^2 - x^2/2s^2 + log(1/sqrt(2 pi s^2))
    cancels catastrophically (terms of size 50-200 summing to O(1) when s is near its minimum 0.1), and d logp / d mu =
    (x - mu)/s^2 reaches 200: ANY float32 evaluation -- the reference's included -- carries
        |d logp| <= sum_j |dlogp/dy_j| |dy_j| + sum_j |dlogp/dx_j| |dx_j| + c u sum|terms|
    with dy, dx its own output / action errors.  Checked here: (1) the head's outputs y and the sampled actions agree with
    float64 to 1e-5 relative / 2e-6 absolute; (2) logp agrees with float64 within that bound evaluated with the MEASURED dy, dx
    (c = 16, u = 2^-24); the same bound is evaluated for the CPU float32 oracle and both are reported."""
    f = lambda t: np.asarray(t.detach().cpu() if isinstance(t, torch.Tensor) else t, np.float64)
    p64 = [(f(w), f(b)) for w, b in params]
    acts, _ = ppo._fwd64(p64, f(obs), out_tanh=True)
    y = acts[-1]
    k = y.shape[1] // 2
    m, b = nets.var_map(*var)
    mu, sd = y[:, :k], y[:, k:] * m + b
    x = np.clip(mu + sd * f(eps), -1.0, 1.0)
    terms = (np.abs(mu * mu / (2 * sd * sd)) + np.abs(mu * x / (sd * sd)) + np.abs(x * x / (2 * sd * sd))
             + np.abs(np.log(1.0 / np.sqrt(2 * np.pi * sd * sd)))).sum(-1)
    logp = (-(mu * mu) / (2 * sd * sd) + mu * x / (sd * sd) - x * x / (2 * sd * sd) + np.log(1.0 / np.sqrt(2 * np.pi * sd * sd))).sum(-1)
    d_y = np.concatenate([np.abs((x - mu) / (sd * sd)), np.abs(((x - mu) ** 2 / sd ** 3 - 1.0 / sd) * m)], 1)
    d_x = np.abs((mu - x) / (sd * sd))
    with torch.no_grad():
        y_cpu = nets.mlp(params, obs, out_act="tanh")
        a_cpu, lp_cpu = nets.gauss_sample(y_cpu[:, :k], y_cpu[:, k:] * m + b, torch.as_tensor(np.asarray(eps, np.float32)))
    res = {}
    for who, yy, aa, ll in (("hip", f(y_hip)[:, :2 * k], f(act_hip), f(logp_hip)), ("cpu", f(y_cpu), f(a_cpu), f(lp_cpu))):
        dy, dx = np.abs(yy - y), np.abs(aa - x)
        assert (dy <= 1e-5 * np.abs(y) + 2e-6).all(), (who, "head outputs", float(dy.max()))
        assert (dx <= 1e-5 * np.abs(x) + 2e-6).all(), (who, "actions", float(dx.max()))
        assert ((np.abs(aa) == 1.0) == (np.abs(x) == 1.0)).mean() > 0.999   # the clamp (quirk Q9) hits the same elements
        bound = (d_y * dy).sum(-1) + (d_x * dx).sum(-1) + 16 * U32 * terms
        err = np.abs(ll - logp)
        res[who] = (float(err.max()), float((err / bound).max()), float(bound.max()))
    print(f"[fp64 gate] {label} Gaussian logp: max|err| HIP {res['hip'][0]:.1e} (= {res['hip'][1]:.2f} of its derived bound, bound max "
          f"{res['hip'][2]:.1e}); CPU fp32 oracle {res['cpu'][0]:.1e} (= {res['cpu'][1]:.2f} of its bound)")
    assert res["hip"][1] <= 1.0, res
    return res


def without_edge_rows(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, grads_pol, masks, edge, w):
    """The policy gradient of an implementation with the contributions of the clip-edge rows -- the only part a float32 rounding of
    the ratio can legitimately switch on or off -- taken out: grad - sum_i w_i c_i, where c_i is row i's full-branch contribution
    (float64, oracle/ppo.py::minibatch_analytic under the implementation's own ReLU masks: the gradient is linear in the branch
    weights) and w_i the branch weight that implementation took for it (read off its output-layer gradient by the gate).  What is
    left depends on no knife-edge decision, so two implementations can be compared on it DIRECTLY."""
    n = np.asarray(obs).shape[0]
    mp, mv = masks
    out = [[np.asarray(gw, np.float64).copy(), np.asarray(gb, np.float64).copy()] for gw, gb in grads_pol]
    if len(edge) == 0:
        return out
    contrib = _edge_contributions(head, pol, val, obs, acts, old, adv, tgt, clip, ent, mb_ratio, var, mp, mv, edge, n)
    for i, c in zip(edge, contrib):
        if w[i] == 0:
            continue
        for l, (cw, cb) in enumerate(c):
            out[l][0] -= w[i] * cw
            out[l][1] -= w[i] * cb
    return out
