"""ctypes binding of librlppo.so (C ABI: include/rlppo.h).

The HIP library is the product: there is no PyTorch/CPU fallback.  `lib()` raises `NativeLibraryMissing`
when the shared object has not been built (python -c "import __graft_entry__ as g; g.build()"), and every
entry point's non-zero return code is turned into `RuntimeError(rlppo_last_error())`, which lands in the
reference-compatible error handler of `Learner.learn` (reference: rlgym_ppo/learner.py:224-238).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLPPO_LIB") or os.path.join(HERE, "librlppo.so")  # RLPPO_LIB: an alternative build (A/B of compile-time variants)
ABI_VERSION = 6
COMM_ID_BYTES = 128  # RLPPO_COMM_ID_BYTES
MAX_LAYERS = 16
N_STATS = 8
OPT_SYNC_BYTES = 16384  # RLPPO_OPT_SYNC_BYTES
OPT_SYNC_TIMEOUT_WORD = 2  # uint32 index of the barrier-timeout counter in the sync block
REPORT_WS_BYTES = 4096  # RLPPO_REPORT_WS_BYTES
REPORT_OUT_DOUBLES = N_STATS + 4  # RLPPO_REPORT_OUT_DOUBLES
EXP_LINK_HEADER = 64  # RLPPO_EXP_LINK_HEADER
MAX_SLOTS = 8
STAT_ENTROPY, STAT_KL, STAT_VLOSS, STAT_CLIPFRAC, STAT_PLOSS = 0, 1, 2, 3, 4
STAT_PASSES = 7  # host-side: passes of rlppo_ppo_minibatch behind the sums above (summed over ranks with them)
HEAD_DISCRETE, HEAD_MULTIDISCRETE, HEAD_GAUSSIAN = 0, 1, 2
PRECISION_DEFAULT, PRECISION_FP32, PRECISION_BF16, PRECISION_X3 = 0, 1, 2, 3  # rlppo_minibatch_args.precision (1 + mode)


class NativeLibraryMissing(RuntimeError):
    pass


class OptNet(ctypes.Structure):
    """struct rlppo_opt_net (include/rlppo.h)."""
    _fields_ = [
        ("dims", POINTER(c_int32)), ("n_layers", c_int32),
        ("params", c_void_p), ("grads", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
        ("packed", c_void_p), ("gnorm2", c_void_p),
        ("max_norm", c_double), ("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
        ("step", c_int64),
    ]


class ActOpts(ctypes.Structure):
    """struct rlppo_act_opts (include/rlppo.h)."""
    _fields_ = [("precision", c_int32), ("done_value", c_uint32), ("done_words", c_void_p), ("noise_ctl", c_void_p)]


class ReportArgs(ctypes.Structure):
    """struct rlppo_report_args (include/rlppo.h)."""
    _fields_ = [
        ("pol_before", c_void_p), ("pol_now", c_void_p), ("n_pol", c_int64), ("val_before", c_void_p), ("val_now", c_void_p), ("n_val", c_int64),
        ("stats", c_void_p), ("add_passes", c_double), ("timeout_word", c_void_p), ("extra", c_void_p), ("out", c_void_p),
        ("done_word", c_void_p), ("done_value", c_uint32), ("ws", c_void_p),
    ]


class TnProduct(ctypes.Structure):
    """struct rlppo_tn_product (include/rlppo.h)."""
    _fields_ = [
        ("dY", c_void_p), ("ldy", c_int64), ("ny_valid", c_int32), ("X", c_void_p), ("ldx", c_int64), ("kx_valid", c_int32),
        ("dW", c_void_p), ("db", c_void_p), ("out", c_int32), ("in_", c_int32), ("rowtab", c_void_p), ("src_rows", c_int64),
    ]


class MinibatchArgs(ctypes.Structure):
    """struct rlppo_minibatch_args (include/rlppo.h)."""
    _fields_ = [
        ("head", c_int32), ("pol_layers", c_int32), ("val_layers", c_int32), ("act_dim", c_int32), ("slot", c_int32), ("precision", c_int32),
        ("pol_dims", POINTER(c_int32)), ("val_dims", POINTER(c_int32)),
        ("pol_packed", c_void_p), ("val_packed", c_void_p),
        ("pol_packed_r", c_void_p), ("val_packed_r", c_void_p), ("pol_wb16", c_void_p), ("val_wb16", c_void_p), ("pol_grad", c_void_p), ("val_grad", c_void_p),
        ("states", c_void_p), ("ld_states", c_int64), ("n_rows", c_int64), ("actions", c_void_p), ("old_logp", c_void_p),
        ("targets", c_void_p), ("advantages", c_void_p), ("idx", c_void_p), ("mb", c_int64),
        ("ring_base", c_int64), ("ring_cap", c_int64),
        ("clip_range", c_float), ("ent_coef", c_float), ("mb_ratio", c_float), ("var_m", c_float), ("var_b", c_float),
        ("stats", c_void_p), ("workspace", c_void_p), ("ws_bytes", c_size_t),
    ]


_P32 = POINTER(c_int32)
_PACT = POINTER(ActOpts)
# name -> (restype, argtypes); kept in one table so tests can check it against the header's declarations
SIGNATURES = {
    "rlppo_abi_version": (c_int32, []),
    "rlppo_last_error": (c_char_p, []),
    "rlppo_build_id": (c_char_p, []),
    "rlppo_padded_width": (c_int64, [c_int64]),
    "rlppo_padded_out": (c_int64, [c_int64]),
    "rlppo_packed_floats": (c_int64, [_P32, c_int32]),
    "rlppo_flat_floats": (c_int64, [_P32, c_int32]),
    "rlppo_net_pack": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p]),
    "rlppo_pad_rows": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int32,
                                 c_float, c_float]),
    "rlppo_pad_rows_per_feature": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                             c_void_p]),
    "rlppo_forward_workspace_bytes": (c_size_t, [_P32, c_int32, c_int64]),
    "rlppo_mlp_forward": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p,
                                    c_int64, c_void_p, c_size_t, _PACT]),
    "rlppo_discrete_act": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, _PACT]),
    "rlppo_discrete_step_workspace_bytes": (c_size_t, [_P32, c_int32, c_int64]),
    "rlppo_discrete_step": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32, c_float, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, _PACT]),
    "rlppo_discrete_probs": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p,
                                       c_int64, c_void_p, c_void_p, c_size_t, _PACT]),
    "rlppo_categorical_select": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "rlppo_gaussian_act": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_float,
                                     c_float, c_void_p, c_void_p, c_void_p, c_size_t, _PACT]),
    "rlppo_multidiscrete_act": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_size_t, _PACT]),
    "rlppo_act_done_words": (c_int64, [c_int64]),
    "rlppo_host_wait_words": (c_int32, [c_void_p, c_int64, c_uint32, c_int64]),
    "rlppo_discrete_step_one_launch": (c_int32, [_P32, c_int32, c_int64, _PACT]),
    "rlppo_host_window_alloc": (c_int32, [c_size_t, POINTER(c_void_p)]),
    "rlppo_host_window_free": (c_int32, [c_void_p]),
    "rlppo_host_window_flush": (c_int32, [c_void_p]),
    "rlppo_host_push": (c_int32, [c_void_p, c_void_p, c_size_t, c_void_p, c_uint32]),
    "rlppo_host_stage_call": (c_int32, [c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_size_t]),
    "rlppo_host_stage_rows": (c_int32, [c_void_p, c_uint32, c_uint32, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_int64]),
    "rlppo_selection_epoch_ptr": (c_void_p, []),
    "rlppo_gae_workspace_bytes": (c_size_t, [c_int64]),
    "rlppo_gae": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_float,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "rlppo_minibatch_workspace_bytes": (c_size_t, [_P32, c_int32, _P32, c_int32, c_int64]),
    "rlppo_minibatch_workspace_bytes_for": (c_size_t, [_P32, c_int32, _P32, c_int32, c_int64, c_int32]),
    "rlppo_ppo_minibatch": (c_int32, [c_void_p, POINTER(MinibatchArgs)]),
    "rlppo_ppo_join": (c_int32, [c_void_p]),
    "rlppo_clip_adam": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                                  c_double, c_double, c_double, c_int64, c_void_p]),
    "rlppo_clip_adam_pack2": (c_int32, [c_void_p, POINTER(OptNet), POINTER(OptNet), c_void_p]),
    "rlppo_learn_report": (c_int32, [c_void_p, POINTER(ReportArgs)]),
    "rlppo_collector_create": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, POINTER(c_void_p)]),
    "rlppo_collector_destroy": (c_int32, [c_void_p]),
    "rlppo_collector_set_obs": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32]),
    "rlppo_collector_ready": (c_int32, [c_void_p, c_void_p, c_int64, POINTER(c_int64)]),
    "rlppo_collector_send": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "rlppo_collector_collect": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_int32, c_int64,
                                          POINTER(c_int64), POINTER(c_int64)]),
    "rlppo_collector_finish": (c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int32), POINTER(c_int64), POINTER(c_int64)]),
    "rlppo_collector_emit": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlppo_collector_average_reward": (c_int32, [c_void_p, c_int32, POINTER(c_double), POINTER(c_int32)]),
    "rlppo_comm_set_library": (c_int32, [ctypes.c_char_p]),
    "rlppo_comm_unique_id": (c_int32, [c_void_p]),
    "rlppo_comm_init": (c_int32, [c_int32, c_int32, c_void_p]),
    "rlppo_allreduce": (c_int32, [c_void_p, c_void_p, c_int64, c_int32]),
    "rlppo_comm_destroy": (c_int32, []),
    "rlppo_mt19937_seed": (c_int32, [POINTER(c_uint32), c_uint32]),
    "rlppo_mt19937_permutation": (c_int32, [POINTER(c_uint32), c_int64, c_void_p]),
    "rlppo_mt19937_draw_targets": (c_int32, [POINTER(c_uint32), c_int64, c_void_p]),
    "rlppo_apply_swap_targets": (c_int32, [c_int64, c_void_p, c_void_p]),
    "rlppo_torch_cpu_exponential": (c_int32, [c_void_p, c_int64, c_int64, c_double, c_void_p, c_int32]),
    "rlppo_torch_cpu_exponential_words": (c_int32, [c_void_p, c_int64, c_int64, c_void_p]),
    "rlppo_exponential_from_words": (c_int32, [c_void_p, c_int64, c_double, c_void_p]),
    "rlppo_torch_cpu_exponential_chained": (c_int32, [c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rlppo_torch_cpu_exponential_burst": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_void_p, c_int64,
                                                    c_int32, c_int32, c_int32, c_void_p]),
    "rlppo_gather_rows": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64]),
    "rlppo_welford_increment": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_int32]),
    "rlppo_welford_merge": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32]),
    "rlppo_set_inference_precision": (c_int32, [c_int32]),
    "rlppo_set_update_precision": (c_int32, [c_int32]),
    "rlppo_get_update_precision": (c_int32, []),
    "rlppo_wb16_elems": (c_int64, [_P32, c_int32]),
    "rlppo_net_pack_bf16": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p, c_void_p]),
    "rlppo_x3_elems": (c_int64, [_P32, c_int32]),
    "rlppo_net_pack_x3": (c_int32, [c_void_p, _P32, c_int32, c_void_p, c_void_p]),
    "rlppo_dbg_pack_x3": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "rlppo_dbg_gemm_nt_x3": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
                                        c_void_p]),
    "rlppo_dbg_set": (c_int32, [c_int32, c_int32]),
    "rlppo_selection_epoch": (c_int64, []),
    "rlppo_dbg_counter": (c_int64, [c_int32]),
    "rlppo_dbg_gemm_nt": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                    c_int64, c_int32, c_int32, c_int32]),
    "rlppo_dbg_gemm_nt_bits_bytes": (c_size_t, [c_int64, c_int32]),
    "rlppo_dbg_gemm_nt_bits": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int32, c_int32, c_int32, c_void_p]),
    "rlppo_dbg_gemm_nt_b16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                        c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "rlppo_dbg_gemm_tn_b16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                        c_int32, c_int64, c_void_p, c_size_t]),
    "rlppo_dbg_thin_head_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int64]),
    "rlppo_dbg_thin_head_b16": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                          c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p, c_size_t]),
    "rlppo_dbg_gemm_tn_group_workspace_bytes": (c_size_t, [POINTER(TnProduct), c_int32, c_int64]),
    "rlppo_dbg_gemm_tn_group": (c_int32, [c_void_p, POINTER(TnProduct), c_int32, c_int64, c_void_p, c_size_t]),
    "rlppo_dbg_gemm_tn_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int64]),
    "rlppo_dbg_gemm_tn": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                                    c_int32, c_int32, c_int64, c_void_p, c_size_t]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found. rlgym_ppo_amd has no CPU/PyTorch fallback: build the HIP library first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or make -C rlgym_ppo_amd/csrc).")
        # Load order: PyTorch's HIP runtime first.  Seen on the GPU box: with librlppo.so loaded into a process before the
        # first torch.cuda call (build() followed by smoke() in one process), every launch of this library failed with
        # hipErrorNoDevice although torch itself worked; with the runtime initialised first it never does.  On a box without
        # a GPU (the build check) there is nothing to initialise.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # noqa: BLE001 -- a CPU-only box: symbols and layouts can still be checked
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == symbol missing == broken build
            fn.restype = res
            fn.argtypes = args
        v = L.rlppo_abi_version()
        if v != ABI_VERSION:
            raise NativeLibraryMissing(f"librlppo.so ABI {v} != expected {ABI_VERSION}: rebuild it")
        # RLPPO_TUNE="key=value,key=value": tuning switches of rlppo_dbg_set (A/B measurements, profiling)
        for item in filter(None, os.environ.get("RLPPO_TUNE", "").split(",")):
            k, v = item.split("=")
            if L.rlppo_dbg_set(int(k), int(v)) != 0:
                raise RuntimeError(f"RLPPO_TUNE: bad switch {item}")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().rlppo_last_error()
        raise RuntimeError(f"librlppo error {rc}: {msg.decode() if msg else '?'}")


def dims_array(dims):
    return (c_int32 * len(dims))(*[int(d) for d in dims])
