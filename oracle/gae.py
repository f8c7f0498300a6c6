"""oracle/gae.py -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Front-end of the C restatement of `compute_gae` (reference: rlgym_ppo/util/torch_functions.py:36-78).
`gae(...)` returns (value_targets f32[N], advantages f32[N], returns f64[N]) like the reference's
(tensor, tensor, list) triple.  mode "f64" = NumPy<2 promotion (what the HIP kernel implements),
mode "np2" = NumPy>=2 promotion (what produced tests/golden/g3_gae.npz); see gae_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gae_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.gae_oracle.restype = ctypes.c_int
        _LIB.gae_oracle.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_long, ctypes.c_double, ctypes.c_double,
                                                            ctypes.c_int, ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 3
    return _LIB


def gae(rews, dones, truncated, values, gamma=0.99, lmbda=0.95, return_std=1, mode="f64"):
    rews = np.ascontiguousarray(rews, np.float32)
    dones = np.ascontiguousarray(dones, np.float32)
    values = np.ascontiguousarray(values, np.float32)
    n = rews.shape[0]
    assert values.shape[0] == n + 1 and dones.shape[0] == n
    truncated = np.asarray(truncated)
    if truncated.dtype == np.float32:
        t32, t64 = np.ascontiguousarray(truncated), None
    else:
        t32, t64 = None, np.ascontiguousarray(truncated, np.float64)
    vt = np.empty(n, np.float32)
    adv = np.empty(n, np.float32)
    ret = np.empty(n, np.float64)
    use_std = return_std is not None
    rc = _lib().gae_oracle(rews.ctypes.data, dones.ctypes.data,
                           t64.ctypes.data if t64 is not None else None,
                           t32.ctypes.data if t32 is not None else None,
                           values.ctypes.data, n, float(gamma), float(lmbda), int(use_std),
                           float(np.float32(return_std)) if use_std else 1.0, {"f64": 0, "np2": 1}[mode],
                           vt.ctypes.data, adv.ctypes.data, ret.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"gae_oracle failed rc={rc}")
    return vt, adv, ret


def gae_python(rews, dones, truncated, values, gamma=0.99, lmbda=0.95, return_std=1):
    """Small-case pure-Python float64 form of the same recurrence written as two affine scans
    x_t = b_t + a_t * x_{t+1} (the shape the HIP kernel parallelises); used to cross-check the C code."""
    n = len(rews)
    a_adv = np.empty(n)
    b_adv = np.empty(n)
    a_ret = np.empty(n)
    for t in range(n):
        nd = float(np.float32(1) - np.float32(dones[t]))
        nt = 1.0 - float(truncated[t])
        if return_std is not None:
            rn = float(np.clip(np.float32(rews[t]) / np.float32(return_std), np.float32(-10), np.float32(10)))
        else:
            rn = float(rews[t])
        b_adv[t] = rn + gamma * float(values[t + 1]) * nd - float(values[t])
        a_adv[t] = gamma * lmbda * nd * nt
        a_ret[t] = gamma * nd * nt
    adv = np.zeros(n + 1)
    ret = np.zeros(n + 1)
    for t in range(n - 1, -1, -1):
        adv[t] = b_adv[t] + a_adv[t] * adv[t + 1]
        ret[t] = float(rews[t]) + a_ret[t] * ret[t + 1]
    vt = (np.asarray(values[:-1], np.float64) + adv[:-1]).astype(np.float32)
    return vt, adv[:-1].astype(np.float32), ret[:-1]
