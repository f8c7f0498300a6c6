"""util.torch_functions -- drop-in for rlgym_ppo/util/torch_functions.py.

compute_gae keeps the reference signature and return convention (torch_functions.py:36-78) but runs librlppo's
two-launch segmented reverse scan on the GPU; `gae_device` is the same kernel on device-resident tensors (what
Learner.add_new_experience uses, so the 512k-step value list never becomes a Python list).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..engine import Workspace, ptr, stream_ptr

_ws = {}


def _workspace(device, nbytes):
    # one workspace per (device, stream): the header's timeout counter is cleared and read per call, so two streams (or two
    # threads on their own streams) running GAE concurrently must not share one (advisor finding, round 3)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _ws:
        _ws[key] = Workspace(device)
    return _ws[key].get(nbytes)


class GAETimeout(RuntimeError):
    """A look-back wait of rlppo_gae gave up: the affected advantages / value targets / returns are NaN (include/rlppo.h)."""


def raise_if_timed_out(timeouts):
    timeouts = int(timeouts)
    if timeouts:
        raise GAETimeout(f"rlppo_gae: {timeouts} look-back wait(s) timed out; the outputs of the affected chunks are NaN "
                         f"and must not be trained on (GPU shared with a long-running kernel, or a defect)")


def gae_device_deferred(rews, dones, truncated, values, gamma=0.99, lmbda=0.95, return_std=1):
    """gae_device without its synchronisation: returns (value_targets, advantages, returns, timeouts) where `timeouts` is a
    one-element int32 DEVICE tensor -- this call's own snapshot of the workspace header's counter, taken on the stream right
    behind the scan -- that the caller MUST hand to raise_if_timed_out() once it has the value on the host (e.g. folded into a
    device-to-host read it performs anyway: Learner.add_new_experience reads it with the first 150 returns).  None under stream
    capture (the stateless two-launch form runs there and no wait exists)."""
    n = rews.shape[0]
    dev = rews.device
    assert values.shape[0] == n + 1 and dones.shape[0] == n and truncated.shape[0] == n
    vt = torch.empty(n, dtype=torch.float32, device=dev)
    adv = torch.empty(n, dtype=torch.float32, device=dev)
    ret = torch.empty(n, dtype=torch.float32, device=dev)
    if n == 0:
        return vt, adv, ret, None
    std = float("nan") if return_std is None else float(np.float32(return_std))
    ws = _workspace(dev, N.lib().rlppo_gae_workspace_bytes(n))
    capturing = torch.cuda.is_current_stream_capturing()
    hdr = ws[:16].view(torch.int32)
    if not capturing:
        hdr.zero_()  # word 1 = look-back waits that timed out (a grown / recycled workspace holds anything)
    N.check(N.lib().rlppo_gae(stream_ptr(), ptr(rews), ptr(dones), ptr(truncated), ptr(values), n, float(gamma),
                              float(lmbda), std, ptr(vt), ptr(adv), ptr(ret), ptr(ws), ws.numel()))
    return vt, adv, ret, (None if capturing else hdr[1:2].clone())


def gae_device(rews, dones, truncated, values, gamma=0.99, lmbda=0.95, return_std=1, check=True):
    """All inputs fp32 device tensors (values has N+1 entries).  Returns (value_targets, advantages, returns) as
    fp32 device tensors.  return_std=None disables reward scaling (torch_functions.py:62-65).

    `check` (default on): the single-launch scan resolves a chunk's carry by waiting, with a bound, on records other
    workgroups publish; if a wait ever gives up the kernel poisons that chunk's outputs with NaN and counts the event in the
    workspace header.  That must never reach the buffer silently (NaN advantages -> Adam -> the weights are gone), so the
    counter is cleared before the launch and read back after it -- one 4-byte read-back, the only synchronisation of this
    call -- and a non-zero count raises GAETimeout for EVERY caller (compute_gae, user code), whatever part of the outputs the
    caller looks at.  check=False leaves the call asynchronous (timing loops); a caller with a device-to-host read of its own
    uses gae_device_deferred and folds the counter into it (Learner.add_new_experience)."""
    vt, adv, ret, timeouts = gae_device_deferred(rews, dones, truncated, values, gamma, lmbda, return_std)
    if check and timeouts is not None:
        raise_if_timed_out(timeouts.item())
    return vt, adv, ret


def compute_gae(rews, dones, truncated, values, gamma=0.99, lmbda=0.95, return_std=1, device=None):
    """Reference signature: sequences in, (value_targets fp32 tensor, advantages fp32 tensor, returns sequence) out.
    The tensors come back on the CPU like the reference's; `returns` is a float32 numpy array (the reference
    returns a list; callers slice and iterate it, learner.py:370-372)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())

    def up(x):
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(device)

    vt, adv, ret = gae_device(up(rews), up(dones), up(truncated), up(values), gamma, lmbda, return_std)
    return vt.cpu(), adv.cpu(), ret.cpu().numpy()


class MapContinuousToAction(nn.Module):
    """Affine map of the tanh outputs' second half onto [range_min, range_max] (torch_functions.py:15-33)."""

    def __init__(self, range_min=0.1, range_max=1):
        super().__init__()
        tanh_range = [-1, 1]
        self.m = (range_max - range_min) / (tanh_range[1] - tanh_range[0])
        self.b = range_min - tanh_range[0] * self.m

    def forward(self, x):
        n = x.shape[-1] // 2
        return x[..., :n], x[..., n:] * self.m + self.b


class MultiDiscreteRolv(nn.Module):
    """8 categoricals over 21 logits, 2-way heads padded with -inf (torch_functions.py:81-122).  Kept for API
    compatibility (get_backprop_data); sampling and the update use the fused kernels."""

    def __init__(self, bins):
        super().__init__()
        self.distribution = None
        self.bins = bins

    def make_distribution(self, logits):
        parts = torch.split(logits, self.bins, dim=-1)
        triplets = torch.stack(parts[:5], dim=-1)
        duets = torch.nn.functional.pad(torch.stack(parts[5:], dim=-1), pad=(0, 0, 0, 1), value=float("-inf"))
        self.distribution = torch.distributions.Categorical(logits=torch.cat((triplets, duets), dim=-1).swapdims(-1, -2))

    def log_prob(self, action):
        return self.distribution.log_prob(action).sum(dim=-1)

    def sample(self):
        return self.distribution.sample()

    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)
