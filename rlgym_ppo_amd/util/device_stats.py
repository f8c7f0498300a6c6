"""Device forms of WelfordRunningStat's two update rules (reference: rlgym_ppo/util/running_stats.py:28-46 and 71-98).

The host object stays the owner of the state (checkpoints read it); these helpers run the update on the GPU in the state's
own dtype -- float32 as constructed, float64 after `from_json` (np.asarray of Python floats, running_stats.py:121-125) --
and write the result back, bit for bit what the host class computes (tests/test_gpu_vector_rollout.py).
"""
import numpy as np
import torch

from .. import _native as N
from ..engine import ptr, stream_ptr


def _state_to_device(stat, device):
    """(mean, m2, is_f64) as flat device tensors in the state's dtype; anything but float64 is treated as float32."""
    f64 = np.asarray(stat.running_mean).dtype == np.float64
    dt = np.float64 if f64 else np.float32
    mean = torch.from_numpy(np.ascontiguousarray(np.asarray(stat.running_mean, dt).reshape(-1))).to(device)
    m2 = torch.from_numpy(np.ascontiguousarray(np.asarray(stat.running_variance, dt).reshape(-1))).to(device)
    return mean, m2, f64


def _state_from_device(stat, mean, m2):
    shape = np.shape(stat.running_mean)
    dt = np.asarray(stat.running_mean).dtype
    stat.running_mean = mean.cpu().numpy().reshape(shape).astype(dt, copy=False)
    stat.running_variance = m2.cpu().numpy().reshape(shape).astype(dt, copy=False)


def increment(stat, samples_dev):
    """stat.increment(samples, n) for n device rows [n, d] (fp32), sample by sample in order."""
    n, d = samples_dev.shape
    if n == 0:
        return
    x = samples_dev if samples_dev.dtype == torch.float32 and samples_dev.is_contiguous() else samples_dev.float().contiguous()
    mean, m2, f64 = _state_to_device(stat, x.device)
    if mean.numel() != d:
        raise ValueError(f"running statistics hold {mean.numel()} features, samples have {d}")
    N.check(N.lib().rlppo_welford_increment(stream_ptr(), ptr(x), x.stride(0), n, d, ptr(mean), ptr(m2), int(stat.count), int(f64)))
    _state_from_device(stat, mean, m2)
    stat.count += n


def merge(stat, serialized_other, device):
    """stat.increment_from_serialized_other(serialized_other) on the device."""
    d = int(np.prod(stat.shape))
    other_count = serialized_other[-1]
    if other_count == 0:
        return
    om = torch.from_numpy(np.asarray(serialized_other[:d], dtype=np.float32)).to(device)
    ov = torch.from_numpy(np.asarray(serialized_other[d:-1], dtype=np.float32)).to(device)
    mean, m2, f64 = _state_to_device(stat, device)
    N.check(N.lib().rlppo_welford_merge(stream_ptr(), d, ptr(mean), ptr(m2), int(stat.count), ptr(om), ptr(ov), int(other_count),
                                        int(f64)))
    _state_from_device(stat, mean, m2)
    stat.count = stat.count + other_count
