// common.hpp -- shared host/device helpers of librlppo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/rlppo.h"

namespace rlppo {

void set_error(const char *fmt, ...);

#define RLPPO_CHECK_ARG(cond, ...)                \
    do {                                          \
        if (!(cond)) {                            \
            rlppo::set_error(__VA_ARGS__);        \
            return RLPPO_ERR_ARG;                 \
        }                                         \
    } while (0)

#define RLPPO_HIP(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            rlppo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                     \
        }                                                                                       \
    } while (0)

#define RLPPO_LAUNCH_CHECK() RLPPO_HIP(hipGetLastError())

// Function attributes and device properties belong to a DEVICE, and one process may drive several (the design is one process per
// GPU, but nothing in the C ABI forbids more): per-kernel set-up state is therefore a bit per device id, not a process-wide flag.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
};
inline int set_dynamic_lds_once(const void *fn, int bytes, PerDeviceOnce &once) {
    int dev = 0;
    RLPPO_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(once.done.load(std::memory_order_acquire) & bit)) {
        RLPPO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        once.done.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}
// compute units of the current device (cached per device id)
inline int device_cu_count(int *cus) {
    static std::atomic<int> by_dev[64];
    int dev = 0;
    RLPPO_HIP(hipGetDevice(&dev));
    int n = by_dev[dev & 63].load(std::memory_order_relaxed);
    if (n == 0) {
        RLPPO_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        by_dev[dev & 63].store(n, std::memory_order_relaxed);
    }
    *cus = n;
    return 0;
}
// hipGetLastError() is per thread and sticky across libraries: an error another library left unread on this thread (seen on
// the GPU box: PyTorch's device probing leaves hipErrorNoDevice behind when this library was loaded before the first
// torch.cuda call) would be reported by the check after our next launch.  Every launch therefore clears the thread's
// error state first, so RLPPO_LAUNCH_CHECK only ever sees the result of the launch in front of it.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)        \
    do {                                                                   \
        (void)hipGetLastError();                                           \
        kernel<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);       \
    } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- packed network layout ------------------------------------------------------------------------
// For layer l (0-based) with logical in_l = dims[l], out_l = dims[l+1]:
//   Pin_l  = padded_width(dims[0]) for l == 0, else Pout_{l-1}
//   Pout_l = padded_out(dims[l+1])
// packed = for each layer: W[Pout][Pin] | WT[Pin][Pout] | b[Pout]   (all zero padded)
struct LayerLayout {
    int in, out;        // logical
    int pin, pout;      // padded
    int64_t off_w, off_wt, off_b;  // float offsets into the packed buffer
    int64_t off_flat_w, off_flat_b;  // float offsets into the flat arena
};

struct NetLayout {
    int n_layers;
    LayerLayout L[RLPPO_MAX_LAYERS];
    int64_t packed_floats;
    int64_t flat_floats;
};

int make_layout(const int32_t *dims, int32_t n_layers, NetLayout *out);

// ---- kernel launchers shared between translation units ------------------------------------------
enum Epilogue { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_TANH = 2, EPI_MASK = 3 };

// gemm.hip ------------------------------------------------------------------------------------------
// C[m][0:N] = epi(A[m][0:K] . B[n][0:K]^T)   (K % 32 == 0, N = padded out in {32,64,96,128,k*128})
int launch_gemm_nt(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                   const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int N, int K, int epi,
                   int bf16_operands = 0);
// the hidden-layer forward that also writes the ReLU bitmask (epi = EPI_BIAS_RELU) / the dX product masked by it (EPI_MASK);
// returns -1 when that form does not apply (width not a multiple of 128)
size_t nt_bits_floats(int64_t M, int N);
// [r3] paired launches: two products of the same shape (a policy and a critic layer of equal widths) in one grid
struct NtAlt {  // second operand set of a paired gemm_nt launch (M, N, K, leading dimensions and row table shared)
    const float *A = nullptr, *B = nullptr, *bias = nullptr;
    float *C = nullptr;
    unsigned long long *bits = nullptr;
    int interleave = 0;  // 1: the two products share the grid's y range (column tiles of both, same XCD back to back) instead of z
};
// [r3] a one-output head folded into the epilogue of the hidden layer that feeds it (the critic's value = h . w + b, value_estimator.py):
// out[row] (compact, zero-filled by the caller) += the row's partial dot product over the workgroup's 128 columns (+ b from the
// first column tile).  With at most two column tiles the sum of the partials does not depend on their order (0 + a + b == 0 + b + a).
struct NtDot {
    const float *w = nullptr, *b = nullptr;  // the head's weight row and its bias (device memory)
    float *out = nullptr;
};
struct TnPair {  // second operand set of a paired gemm_tn launch + reduction (shapes shared; X unused with a row table)
    const float *dY = nullptr, *X = nullptr;
    float *dW = nullptr, *db = nullptr, *ws = nullptr;
};
int launch_gemm_nt_bits(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int N, int K, int epi, unsigned long long *bits, const unsigned *rowtab = nullptr,
                        int64_t src_rows = 0, const NtAlt *alt = nullptr, const NtDot *dot = nullptr /* [2]: per operand set */);
bool nt_gather_ok(int64_t lda, int64_t src_rows, int N, int K);  // can the forward fetch its rows through a row table?
int launch_gemm_nt_bf16(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int N, int nb, int K, int epi);
void set_infer_bf16(int v);
void set_pair_interleave(int v);
void set_nt_skinny(int on);  // rlppo_dbg_set(40): gemm_nt up to 1024 rows as one wave per 16 x 16 output block (1, default) or as 128-row tiles
int get_infer_bf16();
// dW[n][k] += sum_m dY[m][n] X[m][k] ; db[n] += sum_m dY[m][n]: partial tiles in `ws` (>= tn_partial_floats(out, in, M)
// floats) + a fixed-order reduction into the flat arena
size_t tn_partial_floats(int out, int in, int64_t M);
// rowtab: sample m is X[rowtab[m]] (fused minibatch gather)
int launch_gemm_tn(hipStream_t st, const float *dY, int64_t ldy, int ny_valid, const float *X, int64_t ldx, int kx_valid,
                   float *dW, float *db, int out, int in, int64_t M, float *ws, size_t ws_floats, const unsigned *rowtab = nullptr,
                   int64_t src_rows = 0, const TnPair *pair = nullptr);
bool tn_gather_ok(int64_t ldx, int64_t src_rows);
// [r5] ALL weight-gradient products of a pass as ONE launch + ONE reduction (the products of a backward pass feed nothing but the
// optimiser step, so they need not run where autograd would run them): one round of two workgroups per CU whose (product, tile,
// split) items are sized to equal MFMA work, so the partial tiles of the whole pass are what ONE per-layer launch used to write.
struct TnProduct {
    const float *dY = nullptr;
    int64_t ldy = 0;
    int ny_valid = 0;
    const float *X = nullptr;      // [M][ldx], or the experience buffer's state matrix when rowtab != nullptr
    int64_t ldx = 0;
    int kx_valid = 0;
    float *dW = nullptr, *db = nullptr;
    int out = 0, in = 0;
    const unsigned *rowtab = nullptr;
    int64_t src_rows = 0;
};
constexpr int TN_GROUP_MAX = 2 * RLPPO_MAX_LAYERS;
size_t tn_group_floats(const int *outs, const int *ins, int n, int64_t M);  // workspace floats of launch_gemm_tn_group for these shapes
int launch_gemm_tn_group(hipStream_t st, const TnProduct *prods, int n, int64_t M, float *ws, size_t ws_floats);
void set_tn_group_budget(int workgroups);  // rlppo_dbg_set(38): workgroups of a grouped launch (0 = two per CU)

// gemm_b16.hip: the bf16 update precision, both operands bf16 in memory, fp32 accumulate ---------------------------------
// (A: activations / activation gradients, B: rlppo_net_pack_bf16's W or W^T blocks)
// hidden: relu, rounded output as bf16 (Cb) + fp32 (C) + ReLU bitmask; output layer (!hidden): fp32 C, epi bias / bias+tanh
bool nt_b16_ok(int N, int K, bool hidden);
bool nt_split_ok(int N, int K);  // gemm_split.hip [r4]
void set_split_persistent(int v);
int launch_pack_split(hipStream_t st, const float *S, int64_t ld, int R, int Cc, unsigned short *planes);
int launch_gemm_nt_split(hipStream_t st, const float *A, int64_t lda, const unsigned short *planes, const float *bias, float *C, int64_t ldc,
                         int64_t M, int N, int K, int mode, unsigned long long *bits);
int launch_gemm_nt_b16(hipStream_t st, const unsigned short *A, int64_t lda, const unsigned short *B, int64_t ldb, const float *bias,
                       float *C, int64_t ldc, unsigned short *Cb, int64_t ldcb, int64_t M, int N, int K, int epi, int mode,
                       unsigned long long *bits);  // mode: 0 output layer, 1 hidden layer, 2 rounded + masked dX
void set_b16_wide_tiles(int on);  // rlppo_dbg_set(23): 256 x 256 (default) or 128 x 128 tiles in the bf16 hidden / dX products
bool tn_b16_ok(int pout, int pin);
int launch_gemm_tn_b16(hipStream_t st, const unsigned short *dY, int64_t ldy, const unsigned short *X, int64_t ldx, float *dW,
                       float *db, int pout, int pin, int out, int in, int64_t M, float *ws, size_t ws_floats);

// host_rng.cpp
void set_exp_fast_transform(int on);  // rlppo_dbg_set(24): certified vector transform of rlppo_torch_cpu_exponential (1, default) or libm only

// gemv.hip: the critic's one-output head ---------------------------------------------------------------
bool gemv_head_ok(int out, int kp);
int launch_gemv_fwd(hipStream_t st, const float *x, int64_t ldx, const float *w, const float *b, float *y, int64_t ldy, int64_t n, int kp, int pout);
int launch_gemv_fwd_b16(hipStream_t st, const unsigned short *x, int64_t ldx, const float *w, const float *b, float *y, int64_t ldy,
                        int64_t n, int kp, int pout);
// narrow heads (<= 32 outputs) of the bf16 update precision: bf16 activation in, bf16 gradient out (gemv.hip)
bool thin_head_ok(int out, int kp);
size_t thin_dw_ws_floats(int out, int kp, int64_t n);
int launch_thin_dx_b16(hipStream_t st, const float *dy, int64_t ldy, int out, const float *w, int64_t ldw,
                       const unsigned long long *bits, unsigned short *dxb, int64_t ldc, int kp, int64_t n);
int launch_thin_dw_b16(hipStream_t st, const float *dy, int64_t ldy, const unsigned short *xb, int64_t ldx, float *dw, float *db,
                       int out, int in, int kp, int64_t n, float *ws, size_t ws_floats);
int launch_gemv_dx(hipStream_t st, const float *dy, int64_t ldy, const float *w, const float *mask, int64_t ldm, float *dx, int64_t ldc, int kp, int64_t n);
int launch_gemv_dx_bits(hipStream_t st, const float *dy, int64_t ldy, const float *w, const unsigned long long *bits, float *dx,
                        int64_t ldc, int kp, int64_t n);
int launch_gemv_dw(hipStream_t st, const float *dy, int64_t ldy, const float *x, int64_t ldx, float *dw, float *db, int in, int kp, int64_t n,
                   float *ws = nullptr, size_t ws_floats = 0);

// heads.hip: sampling and loss epilogues ---------------------------------------------------------------
struct LossCfg {
    float clip, clip_lo, clip_hi, ent_coef, mb_ratio, inv_mb;
    float var_m, var_b;
    int64_t ring_base, ring_cap;  // ExperienceBuffer ring: logical row i lives at physical row (i + ring_base) mod ring_cap
};
// logical -> physical row of the ring-resident experience (i < cap, base < cap: one conditional subtraction)
__host__ __device__ __forceinline__ int64_t ring_row(int64_t i, int64_t base, int64_t cap) {
    const int64_t r = i + base;
    return r >= cap ? r - cap : r;
}
int launch_discrete_sample_logits(hipStream_t, const float *, int64_t, int64_t, int, const float *, int64_t *, float *, float *);
int launch_discrete_probs(hipStream_t st, const float *logits, int64_t ld, int64_t n, int A, int clamp, float *probs,
                          int64_t ld_p, int64_t *flat_argmax);
int launch_categorical_select(hipStream_t, const float *, int64_t, int64_t, int, const float *, int64_t *, float *);
int launch_gaussian_sample(hipStream_t, const float *, int64_t, int64_t, int, const float *, float, float, float *, float *,
                           unsigned *done_words = nullptr, unsigned done_value = 0);
int launch_multidiscrete_sample(hipStream_t, const float *, int64_t, int64_t, const float *, int64_t *, float *,
                                unsigned *done_words = nullptr, unsigned done_value = 0);
int launch_value_loss(hipStream_t st, float *vout, int64_t ldv, const int64_t *idx, const float *targets, int64_t mb,
                      const LossCfg &cfg, double *stats);
int launch_discrete_loss(hipStream_t, float *, int64_t, int, float *, int64_t, const int64_t *, const float *, const float *,
                         const float *, const float *, int64_t, const LossCfg &, double *);
int launch_gaussian_loss(hipStream_t, float *, int64_t, int, float *, int64_t, const int64_t *, const float *, const float *,
                         const float *, const float *, int64_t, const LossCfg &, double *);
int launch_multidiscrete_loss(hipStream_t, float *, int64_t, float *, int64_t, const int64_t *, const float *, const float *,
                              const float *, const float *, int64_t, const LossCfg &, double *);

// fused_act.hip: the whole rollout step of the discrete policy in one launch ----------------
struct FusedActIO {
    const float *rows = nullptr;  // padded rows [n][ld_rows] ...
    int64_t ld_rows = 0;
    const void *raw = nullptr;    // ... or raw observations [n][ld_raw] (fp32 / fp64) + the standardisation of rlppo_pad_rows(_per_feature)
    int raw_is_f64 = 0, standardize = 0;  // 0 none, 1 scalars (mean0, std0), 2 per-feature vectors
    int64_t ld_raw = 0;
    float mean0 = 0.f, std0 = 1.f;
    const float *mean_v = nullptr, *std_v = nullptr;
    float *rows_out = nullptr;    // optional: the padded rows, [n][ld_rows_out]
    int64_t ld_rows_out = 0;
    const float *noise = nullptr;
    int64_t *actions = nullptr;
    float *actions_f32 = nullptr;  // optional
    float *logp = nullptr, *probs_out = nullptr;
    unsigned *done_words = nullptr;  // [r5] optional, host-visible: word b <- done_value once the outputs of rows 16 b .. 16 b + 15 are visible
    unsigned done_value = 0;
    unsigned *noise_ctl = nullptr;  // [r5] optional control words of noise the host writes after the launch (rlppo_act_opts)
};
bool fused_act_ok(const NetLayout &net);
}  // namespace rlppo
void host_window_register(void *base, size_t bytes, unsigned *hdp_flush);  // host_rng.cpp: the windows of rlppo_host_window_alloc
void host_window_unregister(void *base);
namespace rlppo {
int launch_discrete_act_fused(hipStream_t st, const NetLayout &net, const float *packed, const FusedActIO &io, int64_t n);

// gae.hip ---------------------------------------------------------------------------------------------
size_t gae_workspace_bytes(int64_t n);
void set_gae_algo(int algo);
void set_gae_spin_limit(int v);
void set_gae_oversubscribe(int v);
void set_fused_spin_limit(int v);
void set_fused_test_hold(int v);
int launch_gae(hipStream_t, const float *, const float *, const float *, const float *, int64_t, double, double, float,
               float *, float *, float *, void *, size_t);

// optim.hip -------------------------------------------------------------------------------------------
int launch_pack(hipStream_t, const NetLayout &, const float *, float *);
int launch_pack_bf16(hipStream_t st, const NetLayout &net, const float *flat, float *packed_r, unsigned short *wb16);
int launch_round_rows(hipStream_t st, float *x, unsigned short *xb, int64_t n_elems);
int launch_expand_rows(hipStream_t st, const unsigned short *xb, float *x, int64_t n_elems);
int launch_gather_rows_round(hipStream_t st, const float *src, int64_t ld_src, const int64_t *idx, float *dst, unsigned short *dstb,
                             int width, int64_t n, int64_t ring_base, int64_t ring_cap);
int launch_clip_adam(hipStream_t, float *p, float *g, float *m, float *v, int64_t n, float max_norm, float step_size,
                     float bc2_sqrt, float one_minus_beta1, float beta2, float one_minus_beta2, float eps, double *gnorm2);
int launch_clip_adam_pack2(hipStream_t st, const NetLayout *nets, float *const *p, float *const *g, float *const *m, float *const *v,
                           float *const *packed, double *const *gnorm2, const int64_t *n, const float *max_norm,
                           const float *step_size, const float *bc2_sqrt, const float *omb1, const float *beta2, const float *omb2,
                           const float *eps, void *sync_ws = nullptr);
int launch_welford(hipStream_t st, const float *x, int64_t ld, int64_t n, int d, void *mean, void *m2, long long count0, int state_f64);
int launch_welford_merge(hipStream_t st, int d, void *mean, void *m2, long long count, const float *omean, const float *om2,
                         long long ocount, int state_f64);
struct GatherMeta {  // [r5] the per-row scalars of gather_meta_kernel, gathered by gather_rows_kernel in the same launch
    const float *actions = nullptr, *old_logp = nullptr, *adv = nullptr, *targets = nullptr;
    float *g_act = nullptr, *g_old = nullptr, *g_adv = nullptr, *g_tgt = nullptr, *zero_n = nullptr;
    int act_dim = 0;
};
int launch_gather_rows(hipStream_t st, const float *src, int64_t ld_src, const int64_t *idx, float *dst, int width, int64_t n,
                       int64_t ring_base = 0, int64_t ring_cap = INT64_MAX, const GatherMeta *meta = nullptr);
int launch_gather_meta(hipStream_t st, const int64_t *idx, const float *actions, int act_dim, const float *old_logp,
                       const float *adv, const float *targets, float *g_act, float *g_old, float *g_adv, float *g_tgt, int64_t n,
                       int64_t ring_base, int64_t ring_cap, unsigned *rowtab = nullptr, float *zero_n = nullptr);
int launch_i64_to_f32(hipStream_t st, const int64_t *src, float *dst, int64_t n);
int launch_learn_report(hipStream_t st, const rlppo_report_args &r);  // [r6] the tail of a learn(): norms of the update, statistics out, one completion word
int launch_signal_words(hipStream_t st, unsigned *words, int count, unsigned value);  // words[0..count) <- value, released at system scope
int launch_pad_rows(hipStream_t, const void *, int, int64_t, int64_t, int64_t, float *, int64_t, int, float, float);
int launch_pad_rows_vec(hipStream_t, const void *, int, int64_t, int64_t, int64_t, float *, int64_t, const float *, const float *);

}  // namespace rlppo
