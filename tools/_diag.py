"""ctypes binding of librlppo_diag.so (include/rlppo_diag.h): the measurement probes.  Build it with
`make -C rlgym_ppo_amd/csrc diag`; it is not part of the product library and nothing under rlgym_ppo_amd/ loads it."""
import ctypes
import os
from ctypes import c_int32, c_int64, c_size_t, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "rlgym_ppo_amd", "librlppo_diag.so")
SIGNATURES = {
    "rlppo_dbg_mfma_probe": [c_void_p, c_void_p, c_int32, c_int32, c_void_p],
    "rlppo_dbg_probe2": [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32],
    "rlppo_dbg_probe_ld": [c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_int32, c_void_p],
    "rlppo_dbg_probe_coissue": [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p],
    "rlppo_dbg_gemm_nt_stamped": [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                  c_int32, c_void_p, c_int32],
    "rlppo_dbg_stream_floor": [c_void_p] * 8 + [c_int64, c_int32],
    "rlppo_dbg_gemm_nt_split": [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32],
}


def load():
    import torch
    torch.cuda.init()  # PyTorch's HIP runtime first (same load-order rule as rlgym_ppo_amd/_native.py)
    if not os.path.exists(PATH):
        raise RuntimeError(f"{PATH} not found: make -C rlgym_ppo_amd/csrc diag")
    L = ctypes.CDLL(PATH)
    L.rlppo_diag_last_error.restype = ctypes.c_char_p
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = c_int32, args
    return L


DL = load()


def check(rc):
    if rc != 0:
        raise RuntimeError(f"librlppo_diag error {rc}: {DL.rlppo_diag_last_error().decode()}")
