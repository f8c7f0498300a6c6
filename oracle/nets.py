"""oracle/nets.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

torch-CPU fp32 restatement of the reference's networks and action heads.  A network is a plain list
`[(W0, b0), (W1, b1), ...]` of fp32 tensors (W: [out, in]); nothing here is an nn.Module, so the
arithmetic below is exactly what the HIP kernels have to reproduce, written out op by op.

Reference lines restated:
    MLP bodies ............ discrete_policy.py:22-31, continuous_policy.py:31-41,
                            multi_discrete_policy.py:23-32, value_estimator.py:19-28
    discrete head ......... discrete_policy.py:44-80
    gaussian head ......... continuous_policy.py:43-121, util/torch_functions.py:15-33
    multi-discrete head ... multi_discrete_policy.py:46-89, util/torch_functions.py:81-122
    sampling .............. torch.multinomial(p, 1, True) == argmax(p / q), q ~ Exp(1) drawn with
                            exponential_() from the CPU generator; Normal.sample == mean + std * eps,
                            eps drawn with normal_(0, 1)  (pinned by g1/g9 fixtures)
"""
import math

import numpy as np
import torch

PROB_MIN = 1e-11  # discrete_policy.py:54
MD_BINS = (3, 3, 3, 3, 3, 2, 2, 2)  # multi_discrete_policy.py:20


# ------------------------------------------------------------------------------------------ params
def init_mlp(d_in, hidden, d_out):
    """nn.Linear default init (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for W, same bound for b),
    drawn in construction order from the global CPU generator -- the order the reference's constructors
    consume it (ppo_learner.py:34-53: policy first, then critic)."""
    dims = [int(d_in)] + [int(h) for h in hidden] + [int(d_out)]
    out = []
    for i in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        out.append((lin.weight.detach().clone(), lin.bias.detach().clone()))
    return out


def params_from_state(state, prefix):
    """[(W,b)...] from a fixture / state_dict with keys '<prefix>model.{0,2,..}.{weight,bias}'."""
    idx = sorted({int(k[len(prefix):].split(".")[1]) for k in state if k.startswith(prefix + "model.")})
    return [(torch.as_tensor(np.asarray(state[f"{prefix}model.{i}.weight"]), dtype=torch.float32),
             torch.as_tensor(np.asarray(state[f"{prefix}model.{i}.bias"]), dtype=torch.float32)) for i in idx]


def flatten(params):
    """torch.nn.utils.parameters_to_vector order: W0, b0, W1, b1, ..."""
    return torch.cat([t.reshape(-1) for wb in params for t in wb])


def unflatten(vec, like):
    out, o = [], 0
    for w, b in like:
        nw, nb = w.numel(), b.numel()
        out.append((vec[o:o + nw].view_as(w), vec[o + nw:o + nw + nb].view_as(b)))
        o += nw + nb
    return out


# ----------------------------------------------------------------------------------------- forward
def as_obs(obs):
    """np/any dtype -> fp32 tensor (discrete_policy.py:35-40, value_estimator.py:30-35)."""
    if not isinstance(obs, torch.Tensor):
        obs = torch.as_tensor(np.asarray(obs), dtype=torch.float32)
    return obs


_BF16_OPERANDS = [False]


class bf16_operands:
    """Context manager: inside it every forward product of mlp() rounds both operands to bf16 (the build's optional bf16
    update / rollout precision, BASELINE configs[4]); autograd through it is mixed-precision training as torch writes it: the
    hidden activations are bf16 tensors (`h.bfloat16().float()`), so the gradient with respect to each of them is rounded to bf16
    as well (autograd's cast backward is a cast) and EVERY product of the backward pass -- dX = dY . r(W), dW = dY^T . r(X) --
    multiplies bf16 values with fp32 accumulation; the weights are fp32 master copies whose gradient stays fp32 (_RoundBF16:
    the rounding of a weight has an identity backward, dW and db are never rounded)."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _BF16_OPERANDS[0]
        _BF16_OPERANDS[0] = self.on

    def __exit__(self, *exc):
        _BF16_OPERANDS[0] = self.prev


def mlp(params, x, out_act=None):
    if _BF16_OPERANDS[0]:
        return mlp_bf16_operands(params, x, out_act)
    h = as_obs(x)
    for i, (w, b) in enumerate(params):
        h = torch.nn.functional.linear(h, w, b)
        if i < len(params) - 1:
            h = torch.relu(h)
    if out_act == "tanh":
        h = torch.tanh(h)
    return h


class _RoundBF16(torch.autograd.Function):
    """x -> the nearest bf16 value (round-to-nearest-even), as fp32, with an IDENTITY backward: the rounding of an fp32 MASTER
    WEIGHT (its gradient is accumulated in fp32, unrounded).  Activations use plain `.bfloat16().float()`, whose backward rounds
    the gradient to bf16 -- the gradient of a bf16 tensor."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _LinearSum64(torch.autograd.Function):
    """F.linear whose three products are summed in float64 and rounded ONCE to fp32: the same fp32 function evaluated with a
    different (equally valid) summation.  Used to measure how much of a bf16-forward result is decided by fp32 summation order
    alone (an activation within that noise of a bf16 rounding boundary rounds the other way, and the flip cascades through
    the layers): the floor below which no two implementations of the mode can agree (tests/test_gpu_cfg5.py)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return (x.double() @ w.double().T + b.double()).float()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return (g.double() @ w.double()).float(), (g.double().T @ x.double()).float(), g.double().sum(0).float()


_SUM64 = [False]


class sum64:
    """Context manager: mlp_bf16_operands sums its products in float64 (see _LinearSum64)."""

    def __enter__(self):
        _SUM64[0] = True

    def __exit__(self, *exc):
        _SUM64[0] = False


def mlp_bf16_operands(params, x, out_act=None):
    """The build's optional rollout precision (BASELINE configs[4] "bf16 fwd / fp32 master weights"; the reference has no
    such mode): activations and weights rounded to bf16 (round-to-nearest-even) as they enter each product, fp32
    accumulation, fp32 bias and activation.  Products of two bf16 values are exact in fp32, so only the summation order
    separates this restatement from the MFMA kernel.  Under autograd the activation casts round the gradients that flow back
    through them (see bf16_operands)."""
    h = as_obs(x)
    for i, (w, b) in enumerate(params):
        lin = _LinearSum64.apply if _SUM64[0] else torch.nn.functional.linear
        h = lin(h.bfloat16().float(), _RoundBF16.apply(w), b)
        if i < len(params) - 1:
            h = torch.relu(h)
    if out_act == "tanh":
        h = torch.tanh(h)
    return h


def value_forward(params, obs):
    return mlp(params, obs)  # [n, 1]


# ------------------------------------------------------------------------------------------ discrete
def discrete_probs(params, obs):
    p = torch.softmax(mlp(params, obs), dim=-1)
    return torch.clamp(p, min=PROB_MIN, max=1)


def draw_exp_noise(n, a):
    """The noise torch.multinomial would draw for an [n, a] probability matrix."""
    return torch.empty(n, a).exponential_(1)


def discrete_sample(probs, q):
    act = torch.argmax(probs / q, dim=-1)
    logp = torch.log(probs).gather(-1, act[:, None]).flatten()
    return act, logp


def discrete_backprop(params, obs, acts):
    probs = discrete_probs(params, obs)
    logp_all = torch.log(probs)
    logp = logp_all.gather(-1, acts.long().view(-1, 1))
    ent = -(logp_all * probs).sum(dim=-1)
    return logp.flatten(), ent.mean()


# ------------------------------------------------------------------------------------------ gaussian
def var_map(var_min, var_max):
    m = (var_max - var_min) / 2.0  # torch_functions.py:26-28 with tanh_range [-1, 1]
    return m, var_min + m


def gauss_out(params, obs, var_min=0.1, var_max=1.0):
    y = mlp(params, obs, out_act="tanh")
    k = y.shape[-1] // 2
    m, b = var_map(var_min, var_max)
    return y[..., :k], y[..., k:] * m + b


def gauss_logpdf(x, mean, std):
    msq, ssq, xsq = mean * mean, std * std, x * x
    t1 = -torch.divide(msq, 2 * ssq)
    t2 = torch.divide(mean * x, ssq)
    t3 = -torch.divide(xsq, 2 * ssq)
    t4 = torch.log(1 / torch.sqrt(2 * np.pi * ssq))
    return t1 + t2 + t3 + t4


def gauss_sample(mean, std, eps):
    act = (eps * std + mean).clamp(min=-1, max=1)  # quirk Q9: clamp before log-prob
    return act, gauss_logpdf(act, mean, std).sum(dim=-1)


def gauss_backprop(params, obs, acts, var_min=0.1, var_max=1.0):
    mean, std = gauss_out(params, obs, var_min, var_max)
    logp = gauss_logpdf(acts, mean, std).sum(dim=1)
    ent = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).mean()  # Normal.entropy().mean(), quirk Q8
    return logp, ent


# ------------------------------------------------------------------------------------- multi-discrete
def md_logits3(logits):
    """[n, 21] -> [n, 8, 3] with -inf in the third slot of the three 2-way heads."""
    n = logits.shape[0]
    tri = logits[:, :15].reshape(n, 5, 3)
    duo = logits[:, 15:].reshape(n, 3, 2)
    duo = torch.cat([duo, torch.full((n, 3, 1), float("-inf"))], dim=-1)
    return torch.cat([tri, duo], dim=1)


def md_dist(params, obs):
    l3 = md_logits3(mlp(params, obs))
    lsm = l3 - torch.logsumexp(l3, dim=-1, keepdim=True)  # Categorical(logits=...) normalisation
    return lsm, torch.softmax(l3, dim=-1)


def md_sample(lsm, probs, q):
    n = lsm.shape[0]
    act = torch.argmax(probs.reshape(n * 8, 3) / q, dim=-1).reshape(n, 8)
    return act, lsm.gather(-1, act[..., None]).squeeze(-1).sum(dim=-1)


def md_backprop(params, obs, acts):
    lsm, probs = md_dist(params, obs)
    logp = lsm.gather(-1, acts.long()[..., None]).squeeze(-1).sum(dim=-1)
    ent = -(torch.clamp(lsm, min=torch.finfo(lsm.dtype).min) * probs).sum(-1).sum(-1)
    return logp, ent.mean()


def md_deterministic(params, obs):
    logits = mlp(params, obs)
    out, s = [], 0
    for b in MD_BINS:
        out.append(logits[..., s:s + b].argmax(dim=-1))
        s += b
    return torch.stack(out)  # [8, n] (multi_discrete_policy.py:57-64)


HEADS = {0: "discrete", 1: "multidiscrete", 2: "gaussian"}  # policy_type codes, ppo_learner.py:34-50


def backprop(head, params, obs, acts, var_range=(0.1, 1.0)):
    if head == "discrete":
        return discrete_backprop(params, obs, acts)
    if head == "gaussian":
        return gauss_backprop(params, obs, acts, *var_range)
    return md_backprop(params, obs, acts)
