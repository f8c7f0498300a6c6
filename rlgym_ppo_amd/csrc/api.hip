// api.hip -- the extern "C" surface declared in include/rlppo.h: argument checking, packed-layout bookkeeping and
// the launch sequences (forward, sampling, GAE, one PPO minibatch, clip+Adam).  No device allocation, no
// synchronisation: everything is enqueued on the caller's stream.
#include "build_id.h"
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace rlppo {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Width the first layer's contraction is padded to (zero columns in the packed weights, zero-padded observation rows).  The
// fp32 kernels step K in tiles of 16 and the bf16 kernels in tiles of 64: pad to 16 when that saves at least a tenth of the
// contraction over padding to 64 (107 -> 112 instead of 128: an eighth of the first layer's products, forward and dW, multiplied
// zeros [r3]), else to 64 (231 -> 256: the bf16 update precision keeps its kernels on BASELINE configs[4]).
static int64_t padded_width(int64_t d) {
    const int64_t p16 = round_up(d, 16), p64 = round_up(d, 64);
    return (p64 - p16) * 10 >= p64 ? p16 : p64;
}
static int64_t padded_out(int64_t d) {
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 96) return 96;
    return round_up(d, 128);
}

int make_layout(const int32_t *dims, int32_t n_layers, NetLayout *out) {
    RLPPO_CHECK_ARG(dims != nullptr && n_layers >= 1 && n_layers <= RLPPO_MAX_LAYERS, "network: n_layers=%d not in [1,%d]",
                    n_layers, RLPPO_MAX_LAYERS);
    for (int i = 0; i <= n_layers; ++i) RLPPO_CHECK_ARG(dims[i] >= 1, "network: dims[%d]=%d", i, dims[i]);
    out->n_layers = n_layers;
    int64_t off = 0, flat = 0;
    int pin = (int)padded_width(dims[0]);
    for (int l = 0; l < n_layers; ++l) {
        LayerLayout &L = out->L[l];
        L.in = dims[l];
        L.out = dims[l + 1];
        L.pin = pin;
        L.pout = (int)padded_out(dims[l + 1]);
        L.off_w = off;
        off += (int64_t)L.pout * L.pin;
        L.off_wt = off;
        off += (int64_t)L.pin * L.pout;
        L.off_b = off;
        off += L.pout;
        L.off_flat_w = flat;
        flat += (int64_t)L.out * L.in;
        L.off_flat_b = flat;
        flat += L.out;
        pin = L.pout;
    }
    out->packed_floats = off;
    out->flat_floats = flat;
    return 0;
}

static int max_pout(const NetLayout &net) {
    int m = 0;
    for (int l = 0; l < net.n_layers; ++l) m = net.L[l].pout > m ? net.L[l].pout : m;
    return m;
}

// [r4] The weight image of the split-bf16 update precision (rlppo_set_update_precision(2), csrc/gemm_split.hip): for every hidden
// layer l >= 1 whose shape the split kernels cover, the three bf16 planes of W_l in stage-major order (forward), and for every
// layer l >= 1 whose dX they cover (the output layer included, unless it is a one-output head) the planes of W_l^T.  Offsets in
// bf16 elements; -1 = that product keeps its fp32 kernel.
struct X3Layout {
    int64_t fwd[RLPPO_MAX_LAYERS], dx[RLPPO_MAX_LAYERS], total;
};
static void x3_layout(const NetLayout &net, X3Layout *o) {
    int64_t off = 0;
    const int last = net.n_layers - 1;
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        o->fwd[l] = o->dx[l] = -1;
        if (l >= 1 && l < last && nt_split_ok(L.pout, L.pin)) {
            o->fwd[l] = off;
            off += (int64_t)3 * L.pout * L.pin;
        }
        if (l >= 1 && !(l == last && gemv_head_ok(L.out, L.pin)) && nt_split_ok(L.pin, L.pout)) {
            o->dx[l] = off;
            off += (int64_t)3 * L.pin * L.pout;
        }
    }
    o->total = off;
}

static int g_fold_vhead = 1;  // rlppo_dbg_set(32, 0/1): a one-output head inside the last hidden layer's forward epilogue (update passes)

// Forward pass.  acts[l] receives the output of layer l ([n][pout_l]); for inference the caller passes two
// ping-pong buffers, for training one buffer per layer (they are the saved activations of the backward pass).
// Training only: bits[l] (may be null) receives the ReLU bitmask of hidden layer l and have_bits[l] says whether it was
// written (csrc/gemm.hip, launch_gemm_nt_bits); backward() then masks dX with it instead of re-reading acts[l].
// rowtab != nullptr: `obs` is the experience buffer's state matrix (src_rows rows) and row r of the batch is obs[rowtab[r]]
// (the first layer fetches its rows through the table: nt_gather_ok must hold for it).
static int forward(hipStream_t st, const NetLayout &net, const float *packed, const float *obs, int64_t ld_obs, int64_t n,
                   int out_tanh, float *const *acts, int bf16_operands = 0, unsigned long long *const *bits = nullptr,
                   bool *have_bits = nullptr, const unsigned *rowtab = nullptr, int64_t src_rows = 0, bool *head_folded = nullptr,
                   const unsigned short *x3 = nullptr, bool head_prezeroed = false) {
    X3Layout xl;
    if (x3) x3_layout(net, &xl);
    if (have_bits)
        for (int l = 0; l < net.n_layers; ++l) have_bits[l] = false;
    const float *x = obs;
    int64_t ldx = ld_obs;
    // head_folded != nullptr (update passes): a one-output head may be computed in the epilogue of the hidden layer that feeds it
    // (NtDot, csrc/gemm.hip) -- acts[last] then holds the outputs COMPACT ([n], stride 1) and *head_folded says so
    const int hl = net.n_layers - 1;
    bool fold = head_folded && g_fold_vhead && hl >= 1 && !out_tanh && !bf16_operands && bits && bits[hl - 1] &&
                gemv_head_ok(net.L[hl].out, net.L[hl].pin) && net.L[hl - 1].pout / 128 <= 2;
    if (x3 && hl >= 1 && xl.fwd[hl - 1] >= 0) fold = false;  // (the split forward has no dot-product epilogue: the head keeps its own launch)
    if (head_folded) *head_folded = false;
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        const bool last = l == net.n_layers - 1;
        const int epi = last ? (out_tanh ? EPI_BIAS_TANH : EPI_BIAS) : EPI_BIAS_RELU;
        int rc;
        if (last && head_folded && *head_folded) break;
        if (last && !out_tanh && gemv_head_ok(L.out, L.pin))  // one-output head: matrix-vector kernel (gemv.hip)
            rc = launch_gemv_fwd(st, x, ldx, packed + L.off_w, packed + L.off_b, acts[l], L.pout, n, L.pin, L.pout);
        else {
            rc = -1;
            if (!last && bits && bits[l] && x3 && xl.fwd[l] >= 0) {  // [r4] fp32 data, bf16 MFMA pipe, fp32-grade result
                rc = launch_gemm_nt_split(st, x, ldx, x3 + xl.fwd[l], packed + L.off_b, acts[l], L.pout, n, L.pout, L.pin, 0, bits[l]);
                if (rc == 0) have_bits[l] = true;
            } else if (!last && bits && bits[l] && !bf16_operands) {
                NtDot dots[2];
                const bool fold_here = fold && l == hl - 1;
                if (fold_here) {
                    if (!head_prezeroed) RLPPO_HIP(hipMemsetAsync(acts[hl], 0, (size_t)n * sizeof(float), st));
                    dots[0].w = packed + net.L[hl].off_w;
                    dots[0].b = packed + net.L[hl].off_b;
                    dots[0].out = acts[hl];
                }
                rc = launch_gemm_nt_bits(st, x, ldx, packed + L.off_w, L.pin, packed + L.off_b, acts[l], L.pout, n, L.pout, L.pin,
                                         EPI_BIAS_RELU, bits[l], l == 0 ? rowtab : nullptr, src_rows, nullptr, fold_here ? dots : nullptr);
                if (rc == 0) have_bits[l] = true;
                if (fold_here) *head_folded = rc == 0;  // (rc == -1: the layer has no bitmask form; the values stay zero-filled but unused)
            }
            if (l == 0 && rowtab && rc != 0) {
                if (rc == -1) set_error("forward: the gathered first layer needs the bitmask form");
                return rc == -1 ? RLPPO_ERR_ARG : rc;
            }
            if (rc == -1)
                rc = launch_gemm_nt(st, x, ldx, packed + L.off_w, L.pin, packed + L.off_b, nullptr, 0, acts[l], L.pout, n, L.pout,
                                    L.pin, epi, bf16_operands);
        }
        if (rc) return rc;
        x = acts[l];
        ldx = L.pout;
    }
    return 0;
}

// Forward pass of the bf16 UPDATE precision (rlppo_set_update_precision(1)): every product multiplies bf16-rounded operands
// and accumulates in fp32.  x / xb: the layer input as rounded fp32 and as bf16 (same values); actsb[l] receives the hidden
// output rounded to bf16, bits[l] its ReLU bitmask; acts[l] receives the same values as fp32 only where somebody will read them
// (want_f32[l]: the next layer or its weight gradient takes an fp32 kernel) -- f32_valid[l] reports it; the output layer is
// stored unrounded in fp32.  Layers whose shape gemm_nt_b16_kernel does not cover run the fp32 kernels on the fp32 copies
// (identical products) and are rounded by a separate pass.
static bool b16_next_wants_f32(const NetLayout &net, int l) {  // does the consumer of hidden activation l read it as fp32?
    const int nx = l + 1, last = net.n_layers - 1;
    const LayerLayout &N = net.L[nx];
    if (nx == last) {
        const bool fwd_b16 = gemv_head_ok(N.out, N.pin) || nt_b16_ok(N.pout, N.pin, false);
        return !(fwd_b16 && thin_head_ok(N.out, N.pin));
    }
    return !(nt_b16_ok(N.pout, N.pin, true) && tn_b16_ok(N.pout, N.pin));
}
static int forward_b16(hipStream_t st, const NetLayout &net, const float *packed_r, const unsigned short *wb16, const float *x,
                       const unsigned short *xb, int64_t ldx, int64_t n, int out_tanh, float *const *acts,
                       unsigned short *const *actsb, unsigned long long *const *bits, bool *have_bits, bool *f32_valid) {
    for (int l = 0; l < net.n_layers; ++l) have_bits[l] = f32_valid[l] = false;
    int64_t off16 = 0;
    bool x_f32 = true;  // the gathered states exist in both forms
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        const bool last = l == net.n_layers - 1;
        int rc = 0;
        auto need_x_f32 = [&]() -> int {  // an fp32 kernel is about to read this layer's input
            if (x_f32) return 0;
            x_f32 = f32_valid[l - 1] = true;
            return launch_expand_rows(st, actsb[l - 1], acts[l - 1], n * (int64_t)net.L[l - 1].pout);
        };
        if (last) {
            const int epi = out_tanh ? EPI_BIAS_TANH : EPI_BIAS;
            if (!out_tanh && gemv_head_ok(L.out, L.pin))
                rc = launch_gemv_fwd_b16(st, xb, ldx, packed_r + L.off_w, packed_r + L.off_b, acts[l], L.pout, n, L.pin, L.pout);
            else if (nt_b16_ok(L.pout, L.pin, false))
                rc = launch_gemm_nt_b16(st, xb, ldx, wb16 + off16, L.pin, packed_r + L.off_b, acts[l], L.pout, nullptr, 0, n, L.pout,
                                        L.pin, epi, 0, nullptr);
            else {
                rc = need_x_f32();
                if (rc == 0)
                    rc = launch_gemm_nt(st, x, ldx, packed_r + L.off_w, L.pin, packed_r + L.off_b, nullptr, 0, acts[l], L.pout, n,
                                        L.pout, L.pin, epi);
            }
            f32_valid[l] = true;
        } else if (nt_b16_ok(L.pout, L.pin, true) && bits[l]) {
            const bool want = b16_next_wants_f32(net, l);
            rc = launch_gemm_nt_b16(st, xb, ldx, wb16 + off16, L.pin, packed_r + L.off_b, want ? acts[l] : nullptr, L.pout, actsb[l],
                                    L.pout, n, L.pout, L.pin, EPI_BIAS_RELU, 1, bits[l]);
            have_bits[l] = rc == 0;
            f32_valid[l] = want;
        } else {
            rc = need_x_f32();
            if (rc) return rc;
            rc = -1;
            if (bits[l]) {
                rc = launch_gemm_nt_bits(st, x, ldx, packed_r + L.off_w, L.pin, packed_r + L.off_b, acts[l], L.pout, n, L.pout, L.pin,
                                         EPI_BIAS_RELU, bits[l]);
                have_bits[l] = rc == 0;
            }
            if (rc == -1)
                rc = launch_gemm_nt(st, x, ldx, packed_r + L.off_w, L.pin, packed_r + L.off_b, nullptr, 0, acts[l], L.pout, n, L.pout,
                                    L.pin, EPI_BIAS_RELU);
            if (rc == 0) rc = launch_round_rows(st, acts[l], actsb[l], n * (int64_t)L.pout);
            f32_valid[l] = true;
        }
        if (rc) return rc;
        off16 += (int64_t)L.pout * L.pin;
        x = acts[l];
        xb = actsb[l];
        x_f32 = f32_valid[l];
        ldx = L.pout;
    }
    return 0;
}

static size_t forward_ws_floats(const NetLayout &net, int64_t n) { return (size_t)2 * (size_t)n * (size_t)max_pout(net); }

// runs the net with ping-pong buffers from the workspace and returns the pointer/ld of the last layer's output
static int forward_pingpong(hipStream_t st, const NetLayout &net, const float *packed, const float *obs, int64_t ld_obs,
                            int64_t n, int out_tanh, void *ws, size_t ws_bytes, float *final_out, const float **out,
                            int64_t *ld_out, int bf16_operands) {
    if (ws_bytes < forward_ws_floats(net, n) * sizeof(float)) {
        set_error("forward: workspace %zu < %zu bytes", ws_bytes, forward_ws_floats(net, n) * sizeof(float));
        return RLPPO_ERR_WORKSPACE;
    }
    RLPPO_CHECK_ARG(ld_obs >= net.L[0].pin && ld_obs % 4 == 0, "forward: ld_obs=%ld must be >= %d (zero padded rows)",
                    (long)ld_obs, net.L[0].pin);
    float *b0 = reinterpret_cast<float *>(ws);
    float *b1 = b0 + (size_t)n * max_pout(net);
    float *acts[RLPPO_MAX_LAYERS];
    for (int l = 0; l < net.n_layers; ++l) acts[l] = (l & 1) ? b1 : b0;
    if (final_out) acts[net.n_layers - 1] = final_out;
    int rc = forward(st, net, packed, obs, ld_obs, n, out_tanh, acts, bf16_operands);  // inference only
    if (rc) return rc;
    *out = acts[net.n_layers - 1];
    *ld_out = net.L[net.n_layers - 1].pout;
    return 0;
}

}  // namespace rlppo

using namespace rlppo;

static int g_fused_act = 1;  // rlppo_dbg_set(27, 0/1): rlppo_discrete_act as one fused launch (fused_act.hip)
// rlppo_dbg_counter: which form a call took (tests assert that the kernel they mean to pin is the one that ran)
static std::atomic<long long> g_cnt_fused_act{0}, g_cnt_act_chain{0}, g_cnt_paired_pass{0}, g_cnt_gather_fused_pass{0}, g_cnt_pass{0}, g_cnt_group_dw{0};
// One workgroup per 16 rows and one workgroup per CU (102 KiB of LDS): a launch is rounds of 4096 rows at ~25 us each, whatever
// the round's fill.  Measured (tools/act_kernel_time.py): 64 rows 26 us (chain 70), 4096 rows 29 us (chain 75), 16,384 rows 100 us
// (chain 86): beyond two rounds the layer-by-layer GEMMs, which fill the chip, win.
constexpr int64_t FUSED_ACT_MAX_ROWS = 8192;

// [r5] per-call options of the rollout entry points (rlppo_act_opts): the inference precision of THIS call, and the words the call
// stores into host-visible memory once every output is visible (the host polls them instead of synchronising the stream)
struct ActCtx {
    int bf16 = 0;
    unsigned *done = nullptr;
    unsigned done_value = 0;
    unsigned *noise_ctl = nullptr;
};
// late_noise: the entry point can take its noise while it runs (rlppo_act_opts.noise_ctl; rlppo_discrete_step alone)
static int act_ctx(const rlppo_act_opts *o, ActCtx *c, bool late_noise = false) {
    c->bf16 = get_infer_bf16();
    if (!o) return 0;
    RLPPO_CHECK_ARG(o->precision == RLPPO_PRECISION_DEFAULT || o->precision == RLPPO_PRECISION_FP32 || o->precision == RLPPO_PRECISION_BF16,
                    "act options: inference precision %d (0 = process default, 1 = fp32, 2 = bf16 operands)", o->precision);
    if (o->precision != RLPPO_PRECISION_DEFAULT) c->bf16 = o->precision == RLPPO_PRECISION_BF16;
    c->done = o->done_words;
    c->done_value = o->done_value;
    c->noise_ctl = o->noise_ctl;
    RLPPO_CHECK_ARG(!o->noise_ctl || late_noise, "act options: noise_ctl is an option of rlppo_discrete_step");
    RLPPO_CHECK_ARG(!o->noise_ctl || (o->done_words && !(o->done_value & 0x80000000u)),
                    "act options: noise_ctl needs done_words (a kernel that gives up on the noise reports it there) and done_value < 2^31");
    return 0;
}
// the completion words of a call whose last launch does not write them itself: one more (tiny) launch behind it
static int act_done(hipStream_t st, const ActCtx &c, int64_t n) {
    return c.done ? launch_signal_words(st, c.done, (int)rlppo_act_done_words(n), c.done_value) : 0;
}

extern "C" {

int rlppo_abi_version(void) { return RLPPO_ABI_VERSION; }
const char *rlppo_build_id(void) { return RLPPO_BUILD_ID; }
const char *rlppo_last_error(void) { return g_err; }

int64_t rlppo_padded_width(int64_t d) { return padded_width(d); }
int64_t rlppo_padded_out(int64_t d) { return padded_out(d); }

int64_t rlppo_packed_floats(const int32_t *dims, int32_t n_layers) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return -1;
    return net.packed_floats;
}
int64_t rlppo_flat_floats(const int32_t *dims, int32_t n_layers) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return -1;
    return net.flat_floats;
}

int rlppo_net_pack(void *stream, const int32_t *dims, int32_t n_layers, const float *flat, float *packed) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    RLPPO_CHECK_ARG(flat && packed, "net_pack: null pointer");
    return launch_pack((hipStream_t)stream, net, flat, packed);
}

int rlppo_pad_rows(void *stream, const void *src, int32_t src_is_f64, int64_t n, int64_t d, int64_t ld_src, float *dst,
                   int64_t ld_dst, int32_t standardize, float mean0, float std0) {
    RLPPO_CHECK_ARG(n >= 0 && d >= 1 && ld_src >= d && ld_dst >= d, "pad_rows: n=%ld d=%ld ld_src=%ld ld_dst=%ld", (long)n,
                    (long)d, (long)ld_src, (long)ld_dst);
    RLPPO_CHECK_ARG(n == 0 || (src && dst), "pad_rows: null pointer");
    return launch_pad_rows((hipStream_t)stream, src, src_is_f64, n, d, ld_src, dst, ld_dst, standardize, mean0, std0);
}

int rlppo_pad_rows_per_feature(void *stream, const void *src, int32_t src_is_f64, int64_t n, int64_t d, int64_t ld_src,
                               float *dst, int64_t ld_dst, const float *mean, const float *stdv) {
    RLPPO_CHECK_ARG(n >= 0 && d >= 1 && ld_src >= d && ld_dst >= d, "pad_rows_per_feature: n=%ld d=%ld ld_src=%ld ld_dst=%ld",
                    (long)n, (long)d, (long)ld_src, (long)ld_dst);
    RLPPO_CHECK_ARG(n == 0 || (src && dst && mean && stdv), "pad_rows_per_feature: null pointer");
    return launch_pad_rows_vec((hipStream_t)stream, src, src_is_f64, n, d, ld_src, dst, ld_dst, mean, stdv);
}

size_t rlppo_forward_workspace_bytes(const int32_t *dims, int32_t n_layers, int64_t n) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return 0;
    return forward_ws_floats(net, n > 0 ? n : 0) * sizeof(float) + 256;
}

int rlppo_mlp_forward(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const float *obs,
                      int64_t ld_obs, int64_t n, int32_t out_tanh, float *out, int64_t ld_out, void *workspace,
                      size_t ws_bytes, const rlppo_act_opts *opts) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    ActCtx cx;
    rc = act_ctx(opts, &cx);
    if (rc) return rc;
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && packed && obs && out && workspace, "mlp_forward: bad argument");
    RLPPO_CHECK_ARG(ld_out == net.L[n_layers - 1].pout, "mlp_forward: ld_out=%ld must equal the padded output width %d",
                    (long)ld_out, net.L[n_layers - 1].pout);
    const float *o;
    int64_t ldo;
    rc = forward_pingpong((hipStream_t)stream, net, packed, obs, ld_obs, n, out_tanh, workspace, ws_bytes, out, &o, &ldo, cx.bf16);
    return rc ? rc : act_done((hipStream_t)stream, cx, n);
}

int rlppo_discrete_act(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const float *obs,
                       int64_t ld_obs, int64_t n, const float *noise_q, int64_t *actions, float *logp, float *probs_out,
                       void *workspace, size_t ws_bytes, const rlppo_act_opts *opts) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    ActCtx cx;
    rc = act_ctx(opts, &cx);
    if (rc) return rc;
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && packed && obs && noise_q && actions && logp && workspace, "discrete_act: bad argument");
    // [r3] one launch for the whole step when the network has the form fused_act.hip covers (fp32 inference precision); the
    // layer-by-layer chain below otherwise -- bit-identical results either way
    if (g_fused_act && !cx.bf16 && fused_act_ok(net) && n <= FUSED_ACT_MAX_ROWS && ld_obs >= net.L[0].pin && ld_obs % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(obs) & 15) == 0) {  // (the kernel stages the padded rows with 16-byte loads)
        FusedActIO io;
        io.rows = obs;
        io.ld_rows = ld_obs;
        io.noise = noise_q;
        io.actions = actions;
        io.logp = logp;
        io.probs_out = probs_out;
        io.done_words = cx.done;
        io.done_value = cx.done_value;
        ++g_cnt_fused_act;
        return launch_discrete_act_fused((hipStream_t)stream, net, packed, io, n);
    }
    ++g_cnt_act_chain;
    const float *o;
    int64_t ldo;
    rc = forward_pingpong((hipStream_t)stream, net, packed, obs, ld_obs, n, 0, workspace, ws_bytes, nullptr, &o, &ldo, cx.bf16);
    if (rc) return rc;
    rc = launch_discrete_sample_logits((hipStream_t)stream, o, ldo, n, dims[n_layers], noise_q, actions, logp, probs_out);
    return rc ? rc : act_done((hipStream_t)stream, cx, n);
}

size_t rlppo_discrete_step_workspace_bytes(const int32_t *dims, int32_t n_layers, int64_t n) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return 0;
    n = n > 0 ? n : 0;
    return forward_ws_floats(net, n) * sizeof(float) + ((size_t)n * net.L[0].pin * sizeof(float) + 255) / 256 * 256 + 512;
}

static int g_window_mode = 0;  // rlppo_dbg_set(41, v), test hook: 1 = answer as a device without a large BAR, 2 = windows without a flush register
int rlppo_host_window_alloc(size_t bytes, void **ptr) {
    RLPPO_CHECK_ARG(ptr && bytes > 0, "host_window_alloc: bad argument");
    *ptr = nullptr;
    int dev = 0, large_bar = 0;
    RLPPO_HIP(hipGetDevice(&dev));
    RLPPO_HIP(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev));
    if (g_window_mode == 1) large_bar = 0;
    RLPPO_CHECK_ARG(large_bar, "host_window_alloc: device %d does not expose its memory to the host (no large BAR)", dev);
    RLPPO_HIP(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained));
    RLPPO_HIP(hipMemset(*ptr, 0, bytes));
    RLPPO_HIP(hipDeviceSynchronize());
    // the register whose store flushes the device's host data path: rlppo_host_push / _stage_* store to it behind their bytes
    hipDeviceProp_t prop;
    RLPPO_HIP(hipGetDeviceProperties(&prop, dev));
    host_window_register(*ptr, bytes, g_window_mode == 2 ? nullptr : prop.hdpMemFlushCntl);
    return 0;
}
int rlppo_host_window_free(void *ptr) {
    if (ptr) {
        host_window_unregister(ptr);
        RLPPO_HIP(hipFree(ptr));
    }
    return 0;
}

int rlppo_discrete_step_one_launch(const int32_t *dims, int32_t n_layers, int64_t n, const rlppo_act_opts *opts) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return -1;
    ActCtx cx;
    if (act_ctx(opts, &cx, true)) return -1;
    return g_fused_act && !cx.bf16 && fused_act_ok(net) && n <= FUSED_ACT_MAX_ROWS ? 1 : 0;
}

int rlppo_discrete_step(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const void *obs, int32_t obs_is_f64,
                        int64_t ld_obs, int64_t n, int32_t standardize, float mean0, float std0, const float *mean_v,
                        const float *std_v, const float *noise_q, int64_t *actions, float *actions_f32, float *logp, float *rows_out,
                        int64_t ld_rows_out, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    ActCtx cx;
    rc = act_ctx(opts, &cx, true);
    if (rc) return rc;
    if (n == 0) return 0;
    const int d = net.L[0].in, pin = net.L[0].pin;
    RLPPO_CHECK_ARG(n > 0 && packed && obs && noise_q && actions && logp && workspace, "discrete_step: bad argument");
    RLPPO_CHECK_ARG(ld_obs >= d && standardize >= 0 && standardize <= 2 && (standardize != 2 || (mean_v && std_v)),
                    "discrete_step: ld_obs=%ld standardize=%d", (long)ld_obs, standardize);
    RLPPO_CHECK_ARG(!rows_out || ld_rows_out >= pin, "discrete_step: ld_rows_out=%ld < padded width %d", (long)ld_rows_out, pin);
    hipStream_t st = (hipStream_t)stream;
    if (g_fused_act && !cx.bf16 && fused_act_ok(net) && n <= FUSED_ACT_MAX_ROWS) {
        FusedActIO io;
        io.raw = obs;
        io.raw_is_f64 = obs_is_f64;
        io.ld_raw = ld_obs;
        io.standardize = standardize;
        io.mean0 = mean0;
        io.std0 = std0;
        io.mean_v = mean_v;
        io.std_v = std_v;
        io.rows_out = rows_out;
        io.ld_rows_out = ld_rows_out;
        io.noise = noise_q;
        io.actions = actions;
        io.actions_f32 = actions_f32;
        io.logp = logp;
        io.done_words = cx.done;
        io.done_value = cx.done_value;
        io.noise_ctl = cx.noise_ctl;
        ++g_cnt_fused_act;
        return launch_discrete_act_fused(st, net, packed, io, n);
    }
    RLPPO_CHECK_ARG(!cx.noise_ctl, "discrete_step: noise_ctl needs the one-launch kernel (rlppo_discrete_step_one_launch tells)");
    ++g_cnt_act_chain;
    // the same step launch by launch: pad (+ standardise) into rows_out (or the head of the workspace), forward chain, sample
    float *rows = rows_out;
    int64_t ld_rows = ld_rows_out;
    char *ws = reinterpret_cast<char *>(workspace);
    if (!rows) {
        const size_t need = (size_t)n * pin * sizeof(float);
        RLPPO_CHECK_ARG(ws_bytes >= need + 256, "discrete_step: workspace %zu too small", ws_bytes);
        rows = reinterpret_cast<float *>(ws);
        ld_rows = pin;
        ws += (need + 255) / 256 * 256;
        ws_bytes -= (need + 255) / 256 * 256;
    }
    if (standardize == 2)
        rc = launch_pad_rows_vec(st, obs, obs_is_f64, n, d, ld_obs, rows, ld_rows, mean_v, std_v);
    else
        rc = launch_pad_rows(st, obs, obs_is_f64, n, d, ld_obs, rows, ld_rows, standardize, mean0, std0);
    if (rc) return rc;
    const float *o;
    int64_t ldo;
    rc = forward_pingpong(st, net, packed, rows, ld_rows, n, 0, ws, ws_bytes, nullptr, &o, &ldo, cx.bf16);
    if (rc) return rc;
    rc = launch_discrete_sample_logits(st, o, ldo, n, dims[n_layers], noise_q, actions, logp, nullptr);
    if (rc == 0 && actions_f32) rc = launch_i64_to_f32(st, actions, actions_f32, n);
    return rc ? rc : act_done(st, cx, n);
}

int rlppo_discrete_probs(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const float *obs,
                         int64_t ld_obs, int64_t n, int32_t clamp_probs, float *probs_out, int64_t ld_probs,
                         int64_t *flat_argmax, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    ActCtx cx;
    rc = act_ctx(opts, &cx);
    if (rc) return rc;
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && packed && obs && workspace && (probs_out || flat_argmax), "discrete_probs: bad argument");
    RLPPO_CHECK_ARG(!probs_out || ld_probs >= dims[n_layers], "discrete_probs: ld_probs %lld < n_actions %d", (long long)ld_probs,
                    dims[n_layers]);
    const float *o;
    int64_t ldo;
    rc = forward_pingpong((hipStream_t)stream, net, packed, obs, ld_obs, n, 0, workspace, ws_bytes, nullptr, &o, &ldo, cx.bf16);
    if (rc) return rc;
    rc = launch_discrete_probs((hipStream_t)stream, o, ldo, n, dims[n_layers], clamp_probs != 0, probs_out, ld_probs, flat_argmax);
    return rc ? rc : act_done((hipStream_t)stream, cx, n);
}

int rlppo_categorical_select(void *stream, const float *probs, int64_t ld_p, int64_t n, int32_t n_cat,
                             const float *noise_q, int64_t *actions, float *logp) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && probs && noise_q && actions && logp && n_cat >= 1 && ld_p >= n_cat, "categorical_select: bad argument");
    return launch_categorical_select((hipStream_t)stream, probs, ld_p, n, n_cat, noise_q, actions, logp);
}

int rlppo_gaussian_act(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const float *obs,
                       int64_t ld_obs, int64_t n, const float *noise_eps, float var_m, float var_b, float *actions,
                       float *logp, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    ActCtx cx;
    rc = act_ctx(opts, &cx);
    if (rc) return rc;
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && packed && obs && noise_eps && actions && logp && workspace, "gaussian_act: bad argument");
    RLPPO_CHECK_ARG(dims[n_layers] % 2 == 0, "gaussian_act: output width %d must be 2k", dims[n_layers]);
    const float *o;
    int64_t ldo;
    rc = forward_pingpong((hipStream_t)stream, net, packed, obs, ld_obs, n, 1, workspace, ws_bytes, nullptr, &o, &ldo, cx.bf16);
    if (rc) return rc;
    // (the completion words ride in the sampling kernel: a block of it holds 256 whole rows)
    return launch_gaussian_sample((hipStream_t)stream, o, ldo, n, dims[n_layers] / 2, noise_eps, var_m, var_b, actions, logp, cx.done, cx.done_value);
}

int rlppo_multidiscrete_act(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const float *obs,
                            int64_t ld_obs, int64_t n, const float *noise_q, int64_t *actions, float *logp,
                            void *workspace, size_t ws_bytes, const rlppo_act_opts *opts) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    ActCtx cx;
    rc = act_ctx(opts, &cx);
    if (rc) return rc;
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && packed && obs && noise_q && actions && logp && workspace, "multidiscrete_act: bad argument");
    RLPPO_CHECK_ARG(dims[n_layers] == 21, "multidiscrete_act: output width %d must be 21", dims[n_layers]);
    const float *o;
    int64_t ldo;
    rc = forward_pingpong((hipStream_t)stream, net, packed, obs, ld_obs, n, 0, workspace, ws_bytes, nullptr, &o, &ldo, cx.bf16);
    if (rc) return rc;
    return launch_multidiscrete_sample((hipStream_t)stream, o, ldo, n, noise_q, actions, logp, cx.done, cx.done_value);
}

int64_t rlppo_act_done_words(int64_t n) { return n > 0 ? cdiv(n, 16) : 0; }

size_t rlppo_gae_workspace_bytes(int64_t n) { return gae_workspace_bytes(n) + 256; }

int rlppo_gae(void *stream, const float *rews, const float *dones, const float *truncated, const float *values, int64_t n,
              double gamma, double lmbda, float return_std, float *value_targets, float *advantages, float *returns,
              void *workspace, size_t ws_bytes) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && rews && dones && truncated && values && value_targets && advantages && returns && workspace,
                    "gae: bad argument");
    return launch_gae((hipStream_t)stream, rews, dones, truncated, values, n, gamma, lmbda, return_std, value_targets,
                      advantages, returns, workspace, ws_bytes);
}

// ------------------------------------------------------------------------------------------- PPO minibatch
static int g_two_streams = 1;  // tuning: rlppo_dbg_set(4, 0/1)
static int g_fused_gather = 1;  // rlppo_dbg_set(26, 0/1): first-layer launches fetch their rows through the row table
static int g_update_bf16 = 0;   // rlppo_set_update_precision
// Library-owned streams: slot s > 0 runs its policy chain on main[s]; every slot runs its critic chain on side[s].  One bank per
// device id (streams and events belong to the device that was current when they were made; a process may drive several).
struct SlotBank {
    hipStream_t main[RLPPO_MAX_SLOTS] = {}, side[RLPPO_MAX_SLOTS] = {};
    hipEvent_t ev_fork[RLPPO_MAX_SLOTS] = {}, ev_join[RLPPO_MAX_SLOTS] = {}, ev_slot[RLPPO_MAX_SLOTS] = {}, ev_mid[RLPPO_MAX_SLOTS] = {};
    bool pending[RLPPO_MAX_SLOTS] = {};
};
static SlotBank g_banks[64];
static int slot_bank(SlotBank **bank) {
    int dev = 0;
    RLPPO_HIP(hipGetDevice(&dev));
    RLPPO_CHECK_ARG(dev >= 0 && dev < 64, "device id %d", dev);
    *bank = &g_banks[dev];
    return 0;
}
static int ensure_slot(SlotBank &b, int s) {
    if (!b.side[s]) {
        RLPPO_HIP(hipStreamCreateWithFlags(&b.side[s], hipStreamNonBlocking));
        RLPPO_HIP(hipStreamCreateWithFlags(&b.main[s], hipStreamNonBlocking));
        RLPPO_HIP(hipEventCreateWithFlags(&b.ev_fork[s], hipEventDisableTiming));
        RLPPO_HIP(hipEventCreateWithFlags(&b.ev_join[s], hipEventDisableTiming));
        RLPPO_HIP(hipEventCreateWithFlags(&b.ev_slot[s], hipEventDisableTiming));
        RLPPO_HIP(hipEventCreateWithFlags(&b.ev_mid[s], hipEventDisableTiming));
    }
    return 0;
}
// `to` waits for everything enqueued on `from` so far
static int order_after(hipStream_t to, hipStream_t from, hipEvent_t ev) {
    RLPPO_HIP(hipEventRecord(ev, from));
    RLPPO_HIP(hipStreamWaitEvent(to, ev, 0));
    return 0;
}

// partial-tile workspace of one weight-gradient launch of a net
static size_t tn_layer_floats(const NetLayout &net, int l, int64_t mb, int prec) {
    size_t f = tn_partial_floats(net.L[l].out, net.L[l].in, mb);
    if (net.L[l].out == 1) {  // one-output head: block partials of gemv_dw_kernel (64 rows per block)
        const size_t g = (size_t)cdiv(mb, 64) * (size_t)(net.L[l].pin + 4);  // upper bound: at least 64 rows per block
        if (g > f) f = g;
    }
    if (prec == 1 && l == net.n_layers - 1) {  // narrow head of the bf16 precision: per-lane partials of thin_dw_b16
        const size_t g = thin_dw_ws_floats(net.L[l].out, net.L[l].pin, mb);
        if (g > f) f = g;
    }
    return (f + 3) / 4 * 4;
}
// ... of a net's chain: the launches run one after the other on the net's stream, each followed by its reduction, so one buffer
// of the largest size serves them all.  ([r3] measured: the reductions on streams of their own, one partial buffer per layer --
// a reduction depends on its dW launch only and nothing waits for it before the optimiser step.  The GPU is throughput-bound
// with two chains in flight, so taking them out of the chain saved nothing and their HBM / atomic traffic beside the same
// chain's GEMMs cost 0.6 ms per 10-epoch learn() at one rank and 0.5 ms at the 8-rank share: profiles/r03_ab_update_side_streams.txt.)
static size_t tn_ws_floats(const NetLayout &net, int64_t mb, int prec) {
    size_t m = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        const size_t f = tn_layer_floats(net, l, mb, prec);
        m = f > m ? f : m;
    }
    return m;
}
// [r5] grouped weight gradients (gemm.hip: launch_gemm_tn_group): every dW / db product of a pass that is a GEMM (all layers of both
// networks but a one-output head, which is a matrix-vector kernel) is collected while the chains run their forward / dX launches and
// launched ONCE, after the chains have joined.  The group's partial tiles live in the two chains' partial buffers taken as one region.
static int g_group_dw = 1;  // rlppo_dbg_set(37, 0/1)
struct DwList {
    TnProduct p[2 * RLPPO_MAX_LAYERS];
    int n = 0;
    void add(const float *dY, int64_t ldy, int ny, const float *X, int64_t ldx, int kx, float *dW, float *db, int out, int in,
             const unsigned *rowtab = nullptr, int64_t src_rows = 0) {
        TnProduct &q = p[n++];
        q.dY = dY; q.ldy = ldy; q.ny_valid = ny; q.X = X; q.ldx = ldx; q.kx_valid = kx; q.dW = dW; q.db = db; q.out = out; q.in = in;
        q.rowtab = rowtab; q.src_rows = src_rows;
    }
};
static bool dw_is_gemv(const NetLayout &net, int l) { return l == net.n_layers - 1 && l > 0 && gemv_head_ok(net.L[l].out, net.L[l].pin); }
static size_t tn_region_floats(const NetLayout &pol, const NetLayout &val, int64_t mb, int prec) {
    const size_t chains = tn_ws_floats(pol, mb, prec) + tn_ws_floats(val, mb, prec);
    int outs[2 * RLPPO_MAX_LAYERS], ins[2 * RLPPO_MAX_LAYERS], n = 0;
    for (const NetLayout *net : {&pol, &val})
        for (int l = 0; l < net->n_layers; ++l)
            if (!dw_is_gemv(*net, l)) {
                outs[n] = net->L[l].out;
                ins[n++] = net->L[l].in;
            }
    const size_t group = tn_group_floats(outs, ins, n, mb);
    return chains > group ? chains : group;
}

static size_t train_ws_floats(const NetLayout &pol, const NetLayout &val, int64_t mb, int prec) {
    size_t per_row = 0;
    for (int l = 0; l < pol.n_layers; ++l) per_row += pol.L[l].pout;
    for (int l = 0; l < val.n_layers; ++l) per_row += val.L[l].pout;
    int m = max_pout(pol) > max_pout(val) ? max_pout(pol) : max_pout(val);
    per_row += (size_t)(pol.n_layers - 1 + val.n_layers - 1) * (size_t)m;  // one dX buffer per layer and net
    per_row += (size_t)pol.L[0].pin;                                         // the gathered minibatch states
    per_row += 4 + (size_t)pol.L[pol.n_layers - 1].pout;                      // gathered old log-prob, advantage, target, actions
                                                                             // (act_dim <= the policy's output width) + the row table
    size_t bits = 0;                                                         // ReLU bitmasks of the hidden layers (1/32 of h)
    for (int l = 0; l + 1 < pol.n_layers; ++l) bits += nt_bits_floats(mb, pol.L[l].pout);
    for (int l = 0; l + 1 < val.n_layers; ++l) bits += nt_bits_floats(mb, val.L[l].pout);
    size_t b16 = 0;  // bf16 update precision: the gathered states, every hidden activation and its gradient as bf16 (2 B/element)
    if (prec == 1) {
        size_t el = pol.L[0].pin;  // + per hidden layer: the activation and its gradient
        for (int l = 0; l + 1 < pol.n_layers; ++l) el += 2 * (size_t)pol.L[l].pout;
        for (int l = 0; l + 1 < val.n_layers; ++l) el += 2 * (size_t)val.L[l].pout;
        b16 = (el * (size_t)mb + 1) / 2 + 4;
    }
    return per_row * (size_t)mb + tn_region_floats(pol, val, mb, prec) + bits + b16 + 2;  // + one partial-tile buffer per chain (one region for the grouped launch)
}

// update precision of a call: RLPPO_PRECISION_DEFAULT = what rlppo_set_update_precision chose for the process, else 1 + mode
static int resolve_precision(int32_t precision, int *mode) {
    RLPPO_CHECK_ARG(precision >= 0 && precision <= 3, "update precision %d (0 = process default, 1 = fp32, 2 = bf16 mixed precision, 3 = split-bf16 products)", precision);
    *mode = precision == RLPPO_PRECISION_DEFAULT ? g_update_bf16 : precision - 1;
    return 0;
}
size_t rlppo_minibatch_workspace_bytes_for(const int32_t *pol_dims, int32_t pol_layers, const int32_t *val_dims, int32_t val_layers,
                                           int64_t mb, int32_t precision) {
    NetLayout pol, val;
    int prec = 0;
    if (make_layout(pol_dims, pol_layers, &pol) || make_layout(val_dims, val_layers, &val) || resolve_precision(precision, &prec)) return 0;
    return train_ws_floats(pol, val, mb > 0 ? mb : 0, prec) * sizeof(float) + 256;
}
size_t rlppo_minibatch_workspace_bytes(const int32_t *pol_dims, int32_t pol_layers, const int32_t *val_dims,
                                       int32_t val_layers, int64_t mb) {
    return rlppo_minibatch_workspace_bytes_for(pol_dims, pol_layers, val_dims, val_layers, mb, RLPPO_PRECISION_DEFAULT);
}

// How a chain's first layer finds its rows [r3].
struct ChainCtx {
    const unsigned *rowtab = nullptr;  // fused gather: row r of the pass is src[rowtab[r]] (first-layer dW)
    const float *src = nullptr;
    int64_t ld_src = 0, src_rows = 0;
};

// backward of one net: acts[l] = saved output of layer l, acts[last] holds dL/d(out) on entry; dx[l-1] receives
// dL/d(acts[l-1]) = dY of layer l-1 (one buffer per layer: the dW launches read them later)
static int backward(hipStream_t st, const NetLayout &net, const float *packed, const float *states, int64_t ld_states,
                    int64_t mb, float *const *acts, float *const *dx, float *grad, float *tn_ws,
                    unsigned long long *const *bits, const bool *have_bits, const ChainCtx &cx, bool head_folded = false,
                    const unsigned short *x3 = nullptr, DwList *defer = nullptr) {
    const int last = net.n_layers - 1;
    int rc = 0;
    X3Layout xl;
    if (x3) x3_layout(net, &xl);
    for (int l = last; l >= 0; --l) {
        const LayerLayout &L = net.L[l];
        const float *dY = l == last ? acts[last] : dx[l];
        const int64_t ld_hy = head_folded ? 1 : L.pout;  // stride of a one-output head's dY (compact when it was folded, forward())
        const float *X = l > 0 ? acts[l - 1] : states;
        const int64_t ldx = l > 0 ? net.L[l - 1].pout : ld_states;
        const bool gemv = l == last && l > 0 && gemv_head_ok(L.out, L.pin);  // one-output head (gemv.hip)
        const size_t floats = tn_layer_floats(net, l, mb, 0);  // (fp32 / split-bf16 precisions: the bf16 one has backward_b16)
        if (gemv)
            rc = launch_gemv_dw(st, dY, ld_hy, X, ldx, grad + L.off_flat_w, grad + L.off_flat_b, L.in, L.pin, mb, tn_ws, floats);
        else if (defer && l == 0 && cx.rowtab)  // [r5] collected: launched with every other product of the pass once the chains have joined
            defer->add(dY, L.pout, L.pout, cx.src, cx.ld_src, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in, cx.rowtab, cx.src_rows);
        else if (defer)
            defer->add(dY, L.pout, L.pout, X, ldx, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in);
        else if (l == 0 && cx.rowtab)
            rc = launch_gemm_tn(st, dY, L.pout, L.pout, cx.src, cx.ld_src, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in,
                                mb, tn_ws, floats, cx.rowtab, cx.src_rows);
        else
            rc = launch_gemm_tn(st, dY, L.pout, L.pout, X, ldx, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in, mb,
                                tn_ws, floats);
        if (rc) return rc;
        if (l == 0) break;
        // dX[mb][pin] = (dY[mb][pout] . W[pout][pin]) masked by relu'(acts[l-1]); B operand = W^T [pin][pout].  The mask is the
        // bitmask the forward left when there is one (no re-read of the activation), else the saved activation itself.
        rc = -1;
        if (gemv) {
            if (have_bits[l - 1]) rc = launch_gemv_dx_bits(st, dY, ld_hy, packed + L.off_w, bits[l - 1], dx[l - 1], L.pin, L.pin, mb);
            if (rc == -1) rc = launch_gemv_dx(st, dY, ld_hy, packed + L.off_w, acts[l - 1], L.pin, dx[l - 1], L.pin, L.pin, mb);
        } else {
            if (have_bits[l - 1] && x3 && xl.dx[l] >= 0)  // [r4] split-bf16 dX
                rc = launch_gemm_nt_split(st, dY, L.pout, x3 + xl.dx[l], nullptr, dx[l - 1], L.pin, mb, L.pin, L.pout, 1, bits[l - 1]);
            else if (have_bits[l - 1])
                rc = launch_gemm_nt_bits(st, dY, L.pout, packed + L.off_wt, L.pout, nullptr, dx[l - 1], L.pin, mb, L.pin, L.pout,
                                         EPI_MASK, bits[l - 1]);
            if (rc == -1)
                rc = launch_gemm_nt(st, dY, L.pout, packed + L.off_wt, L.pout, nullptr, acts[l - 1], L.pin, dx[l - 1], L.pin, mb,
                                    L.pin, L.pout, EPI_MASK);
        }
        if (rc) return rc;
    }
    return rc;
}

// Backward of one net in the bf16 update precision: the hidden activations are bf16 tensors,
// so the gradient with respect to each of them is rounded to bf16 (dxb[l], already masked by relu'), and every product
// multiplies bf16 values with fp32 accumulation: dW_l = dxb[l]^T . actsb[l-1] (gemm_tn_b16_kernel), dX = dxb[l] . r(W_l)
// (gemm_nt_b16_kernel, B16_DX).  dW / db accumulate unrounded into the fp32 gradient arena.  The output layer's gradient comes
// from the loss kernel in fp32: narrow heads stream it through thin_dw_b16 / thin_dx_b16 (gemv.hip).  Shapes those kernels do
// not cover take the fp32 kernels on fp32 copies of the same bf16 values (expanded on demand), followed by the same rounding:
// identical mathematics.
static int backward_b16(hipStream_t st, const NetLayout &net, const float *packed_r, const unsigned short *wb16, const float *states,
                        const unsigned short *states_b, int64_t ld_states, int64_t mb, float *const *acts,
                        unsigned short *const *actsb, float *const *dx, unsigned short *const *dxb, float *grad, float *tn_ws,
                        size_t tn_floats, unsigned long long *const *bits, const bool *have_bits, bool *f32_valid) {
    const int last = net.n_layers - 1;
    int64_t first[RLPPO_MAX_LAYERS], total = 0;  // element offsets of the layers' blocks inside the bf16 weight images
    for (int l = 0; l < net.n_layers; ++l) {
        first[l] = total;
        total += (int64_t)net.L[l].pout * net.L[l].pin;
    }
    bool dx_f32[RLPPO_MAX_LAYERS] = {};  // dx[l] holds the fp32 copy of dxb[l]
    int rc = 0;
    auto act_f32 = [&](int l) -> int {  // an fp32 kernel is about to read hidden activation l
        if (l < 0 || f32_valid[l]) return 0;
        f32_valid[l] = true;
        return launch_expand_rows(st, actsb[l], acts[l], mb * (int64_t)net.L[l].pout);
    };
    auto grad_f32 = [&](int l) -> int {  // ... or the gradient of hidden activation l
        if (dx_f32[l]) return 0;
        dx_f32[l] = true;
        return launch_expand_rows(st, dxb[l], dx[l], mb * (int64_t)net.L[l].pout);
    };
    for (int l = last; l >= 0; --l) {
        const LayerLayout &L = net.L[l];
        const float *X = l > 0 ? acts[l - 1] : states;
        const unsigned short *Xb = l > 0 ? actsb[l - 1] : states_b;
        const int64_t ldx = l > 0 ? net.L[l - 1].pout : ld_states;
        const bool gemv = l == last && l > 0 && gemv_head_ok(L.out, L.pin);
        // ---- dW, db
        if (l == last) {
            rc = l > 0 ? launch_thin_dw_b16(st, acts[last], L.pout, Xb, ldx, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in,
                                            L.pin, mb, tn_ws, tn_floats)
                       : -1;
            if (rc == -1) {
                rc = act_f32(l - 1);
                if (rc) return rc;
                if (gemv)
                    rc = launch_gemv_dw(st, acts[last], L.pout, X, ldx, grad + L.off_flat_w, grad + L.off_flat_b, L.in, L.pin, mb,
                                        tn_ws, tn_floats);
                else
                    rc = launch_gemm_tn(st, acts[last], L.pout, L.pout, X, ldx, L.pin, grad + L.off_flat_w, grad + L.off_flat_b,
                                        L.out, L.in, mb, tn_ws, tn_floats);
            }
        } else if (tn_b16_ok(L.pout, L.pin) && ldx % 8 == 0) {
            rc = launch_gemm_tn_b16(st, dxb[l], L.pout, Xb, ldx, grad + L.off_flat_w, grad + L.off_flat_b, L.pout, L.pin, L.out, L.in,
                                    mb, tn_ws, tn_floats);
        } else {
            rc = grad_f32(l);
            if (rc == 0) rc = act_f32(l - 1);
            if (rc) return rc;
            rc = launch_gemm_tn(st, dx[l], L.pout, L.pout, X, ldx, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in, mb,
                                tn_ws, tn_floats);
        }
        if (rc) return rc;
        if (l == 0) break;
        // ---- dX of layer l = the (rounded, masked) gradient of hidden activation l - 1
        if (l < last && have_bits[l - 1] && nt_b16_ok(L.pin, L.pout, true)) {
            rc = launch_gemm_nt_b16(st, dxb[l], L.pout, wb16 + total + first[l], L.pout, nullptr, nullptr, 0, dxb[l - 1], L.pin, mb,
                                    L.pin, L.pout, EPI_MASK, 2, bits[l - 1]);
            if (rc) return rc;
            continue;
        }
        if (l == last && have_bits[l - 1]) {
            rc = launch_thin_dx_b16(st, acts[last], L.pout, L.out, packed_r + L.off_w, L.pin, bits[l - 1], dxb[l - 1], L.pin, L.pin, mb);
            if (rc == 0) continue;
            if (rc != -1) return rc;
        }
        const float *dY = acts[last];
        if (l < last) {
            rc = grad_f32(l);
            if (rc) return rc;
            dY = dx[l];
        }
        rc = -1;
        if (gemv) {
            if (have_bits[l - 1]) rc = launch_gemv_dx_bits(st, dY, L.pout, packed_r + L.off_w, bits[l - 1], dx[l - 1], L.pin, L.pin, mb);
            if (rc == -1) {
                rc = act_f32(l - 1);
                if (rc) return rc;
                rc = launch_gemv_dx(st, dY, L.pout, packed_r + L.off_w, acts[l - 1], L.pin, dx[l - 1], L.pin, L.pin, mb);
            }
        } else {
            if (have_bits[l - 1])
                rc = launch_gemm_nt_bits(st, dY, L.pout, packed_r + L.off_wt, L.pout, nullptr, dx[l - 1], L.pin, mb, L.pin, L.pout,
                                         EPI_MASK, bits[l - 1]);
            if (rc == -1) {
                rc = act_f32(l - 1);
                if (rc) return rc;
                rc = launch_gemm_nt(st, dY, L.pout, packed_r + L.off_wt, L.pout, nullptr, acts[l - 1], L.pin, dx[l - 1], L.pin, mb,
                                    L.pin, L.pout, EPI_MASK);
            }
        }
        if (rc) return rc;
        rc = launch_round_rows(st, dx[l - 1], dxb[l - 1], mb * (int64_t)L.pin);  // the gradient of a bf16 activation is bf16
        if (rc) return rc;
        dx_f32[l - 1] = true;
    }
    return rc;
}

// ---- [r3] paired launches: policy and critic bodies of the same widths (the reference's default: policy_layer_sizes ==
// critic_layer_sizes, learner.py:54-55) run every hidden layer's forward, dX and dW product as ONE launch carrying both networks
// (gemm.hip: NtAlt / TnPair), on one stream; only the two heads (a 90-wide GEMM + loss against a matrix-vector product + loss)
// still fork onto the side stream, where the critic's HBM-bound kernels run beside the policy's MFMA-bound ones.  Same products,
// same summation orders: bit-identical to the two-chain form.  Why: at the 65,536 rows of one rank of an 8-rank job a hidden-layer
// launch is 1024 workgroups = exactly one round of the chip; two such launches on two streams cost two launch boundaries and run
// at the efficiency of the small grid (0.77 of peak isolated against 0.83 at 524,288 rows), 34 launches per optimiser step;
// paired they are 22 launches of two rounds each.  (That reasoning did not survive the measurement at the 8-rank size: see below.)
// Measured (tools/ab_update.py, profiles/r03_ab_update.txt): at one rank (524,288 rows per pass) paired launches take 0.7 % off a
// learn(); at the 65,536 rows of an 8-rank share they ADD 3 % -- with one chain every launch boundary drains the chip, with two
// chains one network's boundary hides under the other's kernel -- so the pairing applies from PAIRED_MIN_ROWS rows per pass up.
static int g_head_order = 1;  // rlppo_dbg_set(31, 0/1): the critic's output-layer backward waits for the policy's loss kernel (both HBM-bound)
static int g_paired = 1;  // rlppo_dbg_set(29, 0 / 1 / 2): never / from PAIRED_MIN_ROWS rows / always (tests)
constexpr int64_t PAIRED_MIN_ROWS = 262144;
constexpr int64_t FUSED_GATHER_MIN_ROWS = 262144;
static bool twin_ok(const NetLayout &p, const NetLayout &v, int64_t mb) {
    if (p.n_layers != v.n_layers || p.n_layers < 2) return false;
    for (int l = 0; l + 1 < p.n_layers; ++l) {
        const LayerLayout &a = p.L[l], &b = v.L[l];
        if (a.in != b.in || a.out != b.out || a.pin != b.pin || a.pout != b.pout) return false;
        if (a.pout % 128 != 0 || a.pin % 16 != 0 || nt_bits_floats(mb, a.pout) == 0) return false;
    }
    return true;
}

// output layer of one net, forward (mirrors forward()'s last iteration)
static int head_forward(hipStream_t st, const NetLayout &net, const float *packed, const float *x, int64_t ldx, int64_t n, int out_tanh,
                        float *out) {
    const LayerLayout &L = net.L[net.n_layers - 1];
    if (!out_tanh && gemv_head_ok(L.out, L.pin)) return launch_gemv_fwd(st, x, ldx, packed + L.off_w, packed + L.off_b, out, L.pout, n, L.pin, L.pout);
    return launch_gemm_nt(st, x, ldx, packed + L.off_w, L.pin, packed + L.off_b, nullptr, 0, out, L.pout, n, L.pout, L.pin,
                          out_tanh ? EPI_BIAS_TANH : EPI_BIAS);
}
// ... and backward: dW / db of the output layer and dX into dx_prev, masked by the last hidden layer's bitmask (backward()'s first iteration)
static int head_backward(hipStream_t st, const NetLayout &net, const float *packed, const float *dY, const float *X, int64_t mb,
                         float *dx_prev, float *grad, float *tn_ws, const unsigned long long *bits_prev, int64_t ld_dy = 0,
                         DwList *defer = nullptr) {
    const int last = net.n_layers - 1;
    const LayerLayout &L = net.L[last];
    const int64_t ldx = net.L[last - 1].pout;
    const bool gemv = gemv_head_ok(L.out, L.pin);
    const size_t floats = tn_layer_floats(net, last, mb, 0);
    if (!ld_dy) ld_dy = L.pout;  // (a one-output head folded into the last hidden layer's epilogue keeps its outputs compact: 1)
    int rc = 0;
    if (gemv) rc = launch_gemv_dw(st, dY, ld_dy, X, ldx, grad + L.off_flat_w, grad + L.off_flat_b, L.in, L.pin, mb, tn_ws, floats);
    else if (defer) defer->add(dY, L.pout, L.pout, X, ldx, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in);
    else rc = launch_gemm_tn(st, dY, L.pout, L.pout, X, ldx, L.pin, grad + L.off_flat_w, grad + L.off_flat_b, L.out, L.in, mb, tn_ws, floats);
    if (rc) return rc;
    if (gemv) return launch_gemv_dx_bits(st, dY, ld_dy, packed + L.off_w, bits_prev, dx_prev, L.pin, L.pin, mb);
    rc = launch_gemm_nt_bits(st, dY, L.pout, packed + L.off_wt, L.pout, nullptr, dx_prev, L.pin, mb, L.pin, L.pout, EPI_MASK,
                             const_cast<unsigned long long *>(bits_prev));
    if (rc == -1) {
        set_error("paired pass: the output layer's dX needs the bitmask form");
        return RLPPO_ERR_ARG;
    }
    return rc;
}

int rlppo_ppo_minibatch(void *stream, const rlppo_minibatch_args *a) {
    RLPPO_CHECK_ARG(a != nullptr, "ppo_minibatch: null args");
    NetLayout pol, val;
    int rc = make_layout(a->pol_dims, a->pol_layers, &pol);
    if (rc) return rc;
    rc = make_layout(a->val_dims, a->val_layers, &val);
    if (rc) return rc;
    const int64_t mb = a->mb;
    if (mb == 0) return 0;
    int prec = 0;  // [r5] the precision is an argument of the call (two learners of one process may differ); the process-wide switch is its default
    rc = resolve_precision(a->precision, &prec);
    if (rc) return rc;
    RLPPO_CHECK_ARG(mb > 0 && a->pol_packed && a->val_packed && a->pol_grad && a->val_grad && a->states && a->actions &&
                        a->old_logp && a->targets && a->advantages && a->idx && a->stats && a->workspace,
                    "ppo_minibatch: null pointer");
    RLPPO_CHECK_ARG(val.L[val.n_layers - 1].out == 1, "ppo_minibatch: critic must have one output");
    RLPPO_CHECK_ARG(pol.L[0].in == val.L[0].in, "ppo_minibatch: policy and critic observe different sizes");
    RLPPO_CHECK_ARG(a->ld_states >= pol.L[0].pin && a->ld_states % 4 == 0, "ppo_minibatch: ld_states=%ld < %d",
                    (long)a->ld_states, pol.L[0].pin);
    if (a->ws_bytes < train_ws_floats(pol, val, mb, prec) * sizeof(float)) {
        set_error("ppo_minibatch: workspace %zu < %zu bytes", a->ws_bytes, train_ws_floats(pol, val, mb, prec) * sizeof(float));
        return RLPPO_ERR_WORKSPACE;
    }
    const int n_out = pol.L[pol.n_layers - 1].out;
    if (a->head == RLPPO_HEAD_DISCRETE) RLPPO_CHECK_ARG(a->act_dim == 1, "discrete head: act_dim must be 1");
    if (a->head == RLPPO_HEAD_GAUSSIAN)
        RLPPO_CHECK_ARG(n_out % 2 == 0 && a->act_dim == n_out / 2, "gaussian head: act_dim=%d, outputs=%d", a->act_dim, n_out);
    if (a->head == RLPPO_HEAD_MULTIDISCRETE)
        RLPPO_CHECK_ARG(n_out == 21 && a->act_dim == 8, "multi-discrete head: needs 21 outputs and act_dim 8");

    // ring-resident experience: ring_cap == 0 means "not a ring" (logical row == physical row)
    const int64_t ring_cap = a->ring_cap > 0 ? a->ring_cap : INT64_MAX, ring_base = a->ring_cap > 0 ? a->ring_base : 0;
    RLPPO_CHECK_ARG(ring_base >= 0 && ring_base < ring_cap, "ppo_minibatch: ring_base=%ld not in [0, ring_cap=%ld)", (long)a->ring_base,
                    (long)a->ring_cap);
    const int slot = a->slot;
    RLPPO_CHECK_ARG(slot >= 0 && slot < RLPPO_MAX_SLOTS, "ppo_minibatch: slot %d not in [0, %d)", slot, RLPPO_MAX_SLOTS);
    SlotBank *bank = nullptr;
    rc = slot_bank(&bank);
    if (rc) return rc;
    SlotBank &bk = *bank;
    rc = ensure_slot(bk, slot);
    if (rc) return rc;
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t st = caller;
    if (slot > 0) {  // this minibatch's chains run beside the caller's stream; rlppo_ppo_join() brings them back
        st = bk.main[slot];
        rc = order_after(st, caller, bk.ev_slot[slot]);
        if (rc) return rc;
        bk.pending[slot] = true;
    }
    float *w = reinterpret_cast<float *>(a->workspace);
    float *pact[RLPPO_MAX_LAYERS], *vact[RLPPO_MAX_LAYERS];
    for (int l = 0; l < pol.n_layers; ++l) {
        pact[l] = w;
        w += (size_t)mb * pol.L[l].pout;
    }
    for (int l = 0; l < val.n_layers; ++l) {
        vact[l] = w;
        w += (size_t)mb * val.L[l].pout;
    }
    const int m = max_pout(pol) > max_pout(val) ? max_pout(pol) : max_pout(val);
    float *pdx[RLPPO_MAX_LAYERS], *vdx[RLPPO_MAX_LAYERS];
    for (int l = 0; l + 1 < pol.n_layers; ++l) {
        pdx[l] = w;
        w += (size_t)mb * m;
    }
    for (int l = 0; l + 1 < val.n_layers; ++l) {
        vdx[l] = w;
        w += (size_t)mb * m;
    }

    float *pol_tn_ws = w;  // partial dW tiles, one buffer per chain; the grouped launch takes the whole region
    float *val_tn_ws = w + tn_ws_floats(pol, mb, prec);
    const size_t tn_region = tn_region_floats(pol, val, mb, prec);
    w += tn_region;
    DwList dws;
    DwList *defer = (g_group_dw && prec != 1) ? &dws : nullptr;
    // ReLU bitmasks of the hidden layers, 8-byte aligned (the workspace base is 256-byte aligned by contract of the host)
    if ((reinterpret_cast<uintptr_t>(w) & 7) != 0) ++w;
    unsigned long long *pbits[RLPPO_MAX_LAYERS] = {}, *vbits[RLPPO_MAX_LAYERS] = {};
    bool phave[RLPPO_MAX_LAYERS] = {}, vhave[RLPPO_MAX_LAYERS] = {};
    bool pf32[RLPPO_MAX_LAYERS] = {}, vf32[RLPPO_MAX_LAYERS] = {};  // bf16 precision: which activations also exist as fp32
    for (int l = 0; l + 1 < pol.n_layers; ++l) {
        const size_t f = nt_bits_floats(mb, pol.L[l].pout);
        pbits[l] = f ? reinterpret_cast<unsigned long long *>(w) : nullptr;
        w += f;
    }
    for (int l = 0; l + 1 < val.n_layers; ++l) {
        const size_t f = nt_bits_floats(mb, val.L[l].pout);
        vbits[l] = f ? reinterpret_cast<unsigned long long *>(w) : nullptr;
        w += f;
    }
    // the minibatch gather (experience_buffer.py:82-87): one pass into the workspace, shared by both nets
    float *const states = w;
    const int64_t ld_states = pol.L[0].pin;
    w += (size_t)mb * pol.L[0].pin;
    float *const wmeta = w;
    w += (size_t)mb * (3 + (size_t)pol.L[pol.n_layers - 1].pout);
    unsigned *const rowtab = reinterpret_cast<unsigned *>(w);  // physical buffer row of every row of the pass (gather_meta_kernel)
    w += (size_t)mb;
    // bf16 update precision: bf16 copies of the gathered rows and of the hidden activations, and the rounded weight images
    const bool b16 = prec == 1;
    const bool x3 = prec == 2;  // [r4] split-bf16 hidden forward / dX (csrc/gemm_split.hip); everything else as fp32
    unsigned short *states_b = nullptr, *pactb[RLPPO_MAX_LAYERS] = {}, *vactb[RLPPO_MAX_LAYERS] = {};
    unsigned short *pdxb[RLPPO_MAX_LAYERS] = {}, *vdxb[RLPPO_MAX_LAYERS] = {};
    if (b16) {
        RLPPO_CHECK_ARG(a->pol_packed_r && a->val_packed_r && a->pol_wb16 && a->val_wb16,
                        "ppo_minibatch: the bf16 update precision needs the rlppo_net_pack_bf16 images of both networks");
        if ((reinterpret_cast<uintptr_t>(w) & 15) != 0) w += 4 - ((reinterpret_cast<uintptr_t>(w) >> 2) & 3);
        unsigned short *hb = reinterpret_cast<unsigned short *>(w);
        states_b = hb;
        hb += (size_t)mb * pol.L[0].pin;
        for (int l = 0; l + 1 < pol.n_layers; ++l) {
            pactb[l] = hb;
            hb += (size_t)mb * pol.L[l].pout;
        }
        for (int l = 0; l + 1 < val.n_layers; ++l) {
            vactb[l] = hb;
            hb += (size_t)mb * val.L[l].pout;
        }
        for (int l = 0; l + 1 < pol.n_layers; ++l) {
            pdxb[l] = hb;
            hb += (size_t)mb * pol.L[l].pout;
        }
        for (int l = 0; l + 1 < val.n_layers; ++l) {
            vdxb[l] = hb;
            hb += (size_t)mb * val.L[l].pout;
        }
        // the fp32 form of the gathered rows only if a first layer takes the fp32 kernels (the predicates of forward_b16 / backward_b16)
        auto lean0 = [](const NetLayout &n) {
            return n.n_layers > 1 && nt_b16_ok(n.L[0].pout, n.L[0].pin, true) && tn_b16_ok(n.L[0].pout, n.L[0].pin);
        };
        rc = launch_gather_rows_round(st, a->states, a->ld_states, a->idx, lean0(pol) && lean0(val) ? nullptr : states, states_b,
                                      pol.L[0].pin, mb, ring_base, ring_cap);
    }
    // [r3] fp32 precision: the four first-layer launches (two forwards, two dW) fetch their rows straight from the experience
    // buffer through the row table (the gather fused into the load stage) when their shapes have that form; the
    // separate gather pass into the workspace remains for the others.
    const int64_t src_rows = a->ring_cap > 0 ? a->ring_cap : a->n_rows;
    auto gather_form = [&](const NetLayout &n) {
        return n.n_layers > 1 && nt_gather_ok(a->ld_states, src_rows, n.L[0].pout, n.L[0].pin) && tn_gather_ok(a->ld_states, src_rows) &&
               !(n.L[0].out > 64 && n.L[0].out <= 96 && !(n.L[0].in > 96 && n.L[0].in <= 112));
    };
    // [r5] ... from FUSED_GATHER_MIN_ROWS rows per pass: with the weight gradients grouped (section 4.6) a 65,536-row pass is 1 % FASTER
    // with the 10 us gather pass and contiguous first-layer operands than with rows fetched through the table inside four of its
    // launches (11.59 against 11.71 ms per 10-epoch learn() at the 8-rank share, tools/ab_update.py, profiles/r05_ab_update_final.txt);
    // at 524,288 rows per pass the two are equal and the fused form saves the 268 MB copy.  rlppo_dbg_set(26, 2): at every size (tests).
    const bool fused_gather = !b16 && (g_fused_gather == 2 || (g_fused_gather == 1 && mb >= FUSED_GATHER_MIN_ROWS)) && src_rows > 0 &&
                              gather_form(pol) && gather_form(val);
    // (the per-row scalars -- old log-prob, advantage, target, actions -- ride in the same launch: wmeta below)
    if (!b16 && !fused_gather) {
        RLPPO_CHECK_ARG(a->act_dim >= 1 && a->act_dim <= pol.L[pol.n_layers - 1].pout, "ppo_minibatch: act_dim=%d", a->act_dim);
        GatherMeta gm;
        gm.actions = a->actions; gm.old_logp = a->old_logp; gm.adv = a->advantages; gm.targets = a->targets;
        gm.g_old = wmeta; gm.g_adv = wmeta + mb; gm.g_tgt = wmeta + 2 * (size_t)mb; gm.g_act = wmeta + 3 * (size_t)mb;  // (= g_old, g_adv, g_tgt, g_act below)
        gm.zero_n = vact[val.n_layers - 1];
        gm.act_dim = a->act_dim;
        rc = launch_gather_rows(st, a->states, a->ld_states, a->idx, states, pol.L[0].pin, mb, ring_base, ring_cap, &gm);
    }
    if (rc) return rc;
    const float *pol_w = b16 ? a->pol_packed_r : a->pol_packed, *val_w = b16 ? a->val_packed_r : a->val_packed;
    // the minibatch's per-row scalars, gathered once (the loss kernels stream them)
    RLPPO_CHECK_ARG(a->act_dim >= 1 && a->act_dim <= pol.L[pol.n_layers - 1].pout, "ppo_minibatch: act_dim=%d", a->act_dim);
    float *g_old = wmeta, *g_adv = wmeta + mb, *g_tgt = wmeta + 2 * (size_t)mb, *g_act = wmeta + 3 * (size_t)mb;
    // ([r5] it also zeroes the first mb floats of the critic's output buffer: a folded value head accumulates into them; when the
    // rows were gathered by a pass of their own, that launch has done all of this already)
    if (b16 || fused_gather)
        rc = launch_gather_meta(st, a->idx, a->actions, a->act_dim, a->old_logp, a->advantages, a->targets, g_act, g_old, g_adv, g_tgt, mb,
                                ring_base, ring_cap, fused_gather ? rowtab : nullptr, vact[val.n_layers - 1]);
    if (rc) return rc;
    // forward of both nets
    // The two networks are independent until the loss epilogue and again after it, so their launch chains run on
    // two streams (the caller's + one library-owned side stream, forked/joined with events: capturable).  Each
    // launch is only 50-100 us long at K <= 256, so letting one chain's kernels fill the CUs that the other chain's
    // ramp-up / tail leaves idle is worth more than any single-kernel tweak.
    hipStream_t side = st;
    if (g_two_streams) {
        side = bk.side[slot];
        rc = order_after(side, st, bk.ev_fork[slot]);
        if (rc) return rc;
    }
    const bool twin = !b16 && !x3 && (g_paired == 2 || (g_paired == 1 && mb >= PAIRED_MIN_ROWS)) && twin_ok(pol, val, mb);
    const unsigned short *pol_x3 = x3 ? reinterpret_cast<const unsigned short *>(a->pol_wb16) : nullptr;
    const unsigned short *val_x3 = x3 ? reinterpret_cast<const unsigned short *>(a->val_wb16) : nullptr;
    if (x3) RLPPO_CHECK_ARG(pol_x3 && val_x3, "ppo_minibatch: the split-bf16 update precision needs the rlppo_net_pack_x3 images of both networks (pol_wb16 / val_wb16)");
    ++g_cnt_pass;
    if (twin) ++g_cnt_paired_pass;
    if (fused_gather) ++g_cnt_gather_fused_pass;
    if (twin) {
        const int H = pol.n_layers - 1;  // hidden layers (the same number in both networks)
        const float *xp = fused_gather ? a->states : states, *xv = xp;
        int64_t ldx = fused_gather ? a->ld_states : ld_states;
        // [r3] The critic's output layer is a dot product per row of activations this launch has in registers: folded into the last
        // hidden layer's forward epilogue (NtDot) it costs no pass over the 4 x 256 B per row the matrix-vector kernel re-read
        // (537 MB per 524,288-row pass, in the stretch of the pass that is HBM-bound).  Values land compact ([mb], stride 1).
        const LayerLayout &Lvh = val.L[H];
        const bool fold_v = g_fold_vhead && gemv_head_ok(Lvh.out, Lvh.pin) && val.L[H - 1].pout / 128 <= 2;
        float *vout = vact[H];
        const int64_t ldv = fold_v ? 1 : Lvh.pout;
        for (int l = 0; l < H; ++l) {
            const LayerLayout &Lp = pol.L[l], &Lv = val.L[l];
            NtAlt alt;
            alt.A = xv;
            alt.B = val_w + Lv.off_w;
            alt.bias = val_w + Lv.off_b;
            alt.C = vact[l];
            alt.bits = vbits[l];
            NtDot dots[2];
            const bool fold_here = fold_v && l == H - 1;
            if (fold_here) {  // (vout was zeroed by gather_meta_kernel)
                dots[1].w = val_w + Lvh.off_w;
                dots[1].out = vout;
                dots[1].b = val_w + Lvh.off_b;
            }
            rc = launch_gemm_nt_bits(st, xp, ldx, pol_w + Lp.off_w, Lp.pin, pol_w + Lp.off_b, pact[l], Lp.pout, mb, Lp.pout, Lp.pin,
                                     EPI_BIAS_RELU, pbits[l], l == 0 && fused_gather ? rowtab : nullptr, src_rows, &alt,
                                     fold_here ? dots : nullptr);
            if (rc) {
                if (rc == -1) set_error("paired pass: a hidden layer does not have the bitmask form");
                return rc == -1 ? RLPPO_ERR_ARG : rc;
            }
            xp = pact[l];
            xv = vact[l];
            ldx = Lp.pout;
        }
        // the two heads: forward, loss, output-layer backward -- critic on the side stream
        hipStream_t hs = st;
        if (g_two_streams) {
            hs = bk.side[slot];
            rc = order_after(hs, st, bk.ev_fork[slot]);
            if (rc) return rc;
        }
        LossCfg cfg;
        cfg.clip = a->clip_range;
        cfg.clip_lo = (float)(1.0 - (double)a->clip_range);
        cfg.clip_hi = (float)(1.0 + (double)a->clip_range);
        cfg.ent_coef = a->ent_coef;
        cfg.mb_ratio = a->mb_ratio;
        cfg.inv_mb = 1.0f / (float)mb;
        cfg.var_m = a->var_m;
        cfg.var_b = a->var_b;
        cfg.ring_base = ring_base;
        cfg.ring_cap = ring_cap;
        float *pout = pact[H];
        const int64_t ldp = pol.L[H].pout;
        if (!fold_v) {
            rc = head_forward(hs, val, val_w, xv, ldx, mb, 0, vout);
            if (rc) return rc;
        }
        rc = launch_value_loss(hs, vout, ldv, nullptr, g_tgt, mb, cfg, a->stats);
        if (rc) return rc;
        // The critic's head kernels are matrix-vector products: HBM-bound, like the policy's loss kernel, unlike the policy head's
        // GEMMs.  Ordered so that the two HBM-bound stretches do not meet: critic forward + loss beside the policy head's forward
        // GEMM, the critic's output-layer backward only after the policy's loss, beside the policy head's dW / dX GEMMs.
        const bool ordered = g_head_order && hs != st;
        if (!ordered) {
            rc = head_backward(hs, val, val_w, vout, xv, mb, vdx[H - 1], a->val_grad, val_tn_ws, vbits[H - 1], ldv, defer);
            if (rc) return rc;
        }
        rc = head_forward(st, pol, pol_w, xp, ldx, mb, a->head == RLPPO_HEAD_GAUSSIAN, pout);
        if (rc) return rc;
        if (a->head == RLPPO_HEAD_DISCRETE)
            rc = launch_discrete_loss(st, pout, ldp, n_out, nullptr, ldv, nullptr, g_act, g_old, g_tgt, g_adv, mb, cfg, a->stats);
        else if (a->head == RLPPO_HEAD_GAUSSIAN)
            rc = launch_gaussian_loss(st, pout, ldp, n_out / 2, nullptr, ldv, nullptr, g_act, g_old, g_tgt, g_adv, mb, cfg, a->stats);
        else if (a->head == RLPPO_HEAD_MULTIDISCRETE)
            rc = launch_multidiscrete_loss(st, pout, ldp, nullptr, ldv, nullptr, g_act, g_old, g_tgt, g_adv, mb, cfg, a->stats);
        else {
            set_error("ppo_minibatch: unknown head %d", a->head);
            rc = RLPPO_ERR_ARG;
        }
        if (rc) return rc;
        if (ordered) {
            rc = order_after(hs, st, bk.ev_mid[slot]);
            if (rc) return rc;
            rc = head_backward(hs, val, val_w, vout, xv, mb, vdx[H - 1], a->val_grad, val_tn_ws, vbits[H - 1], ldv, defer);
            if (rc) return rc;
        }
        rc = head_backward(st, pol, pol_w, pout, xp, mb, pdx[H - 1], a->pol_grad, pol_tn_ws, pbits[H - 1], 0, defer);
        if (rc) return rc;
        if (hs != st) {
            rc = order_after(st, hs, bk.ev_join[slot]);
            if (rc) return rc;
        }
        // hidden layers, backward: dW (+ reduction) of both networks, then dX of both
        for (int l = H - 1; l >= 0; --l) {
            const LayerLayout &Lp = pol.L[l], &Lv = val.L[l];
            const bool g0 = l == 0 && fused_gather;
            const float *Xp = l > 0 ? pact[l - 1] : (g0 ? a->states : states), *Xv = l > 0 ? vact[l - 1] : Xp;
            const int64_t ldxb = l > 0 ? pol.L[l - 1].pout : (g0 ? a->ld_states : ld_states);
            TnPair pr;
            pr.dY = vdx[l];
            pr.X = Xv;
            pr.dW = a->val_grad + Lv.off_flat_w;
            pr.db = a->val_grad + Lv.off_flat_b;
            pr.ws = val_tn_ws;
            if (defer) {
                defer->add(pdx[l], Lp.pout, Lp.pout, Xp, ldxb, Lp.pin, a->pol_grad + Lp.off_flat_w, a->pol_grad + Lp.off_flat_b, Lp.out, Lp.in,
                           g0 ? rowtab : nullptr, g0 ? src_rows : 0);
                defer->add(vdx[l], Lv.pout, Lv.pout, Xv, ldxb, Lv.pin, a->val_grad + Lv.off_flat_w, a->val_grad + Lv.off_flat_b, Lv.out, Lv.in,
                           g0 ? rowtab : nullptr, g0 ? src_rows : 0);
            } else
                rc = launch_gemm_tn(st, pdx[l], Lp.pout, Lp.pout, Xp, ldxb, Lp.pin, a->pol_grad + Lp.off_flat_w, a->pol_grad + Lp.off_flat_b,
                                    Lp.out, Lp.in, mb, pol_tn_ws, tn_layer_floats(pol, l, mb, 0), g0 ? rowtab : nullptr, src_rows, &pr);
            if (rc) return rc;
            if (l == 0) break;
            NtAlt alt;
            alt.A = vdx[l];
            alt.B = val_w + Lv.off_wt;
            alt.C = vdx[l - 1];
            alt.bits = vbits[l - 1];
            rc = launch_gemm_nt_bits(st, pdx[l], Lp.pout, pol_w + Lp.off_wt, Lp.pout, nullptr, pdx[l - 1], Lp.pin, mb, Lp.pin, Lp.pout,
                                     EPI_MASK, pbits[l - 1], nullptr, 0, &alt);
            if (rc) return rc == -1 ? RLPPO_ERR_ARG : rc;
        }
        if (defer && dws.n) {
            ++g_cnt_group_dw;
            return launch_gemm_tn_group(st, dws.p, dws.n, mb, pol_tn_ws, tn_region);
        }
        return 0;
    }
    bool v_folded = false;  // the critic's one-output head was computed in its last hidden layer's epilogue: compact outputs
    if (b16) {
        rc = forward_b16(side, val, val_w, reinterpret_cast<const unsigned short *>(a->val_wb16), states, states_b, ld_states, mb, 0,
                         vact, vactb, vbits, vhave, vf32);
        if (rc) return rc;
        rc = forward_b16(st, pol, pol_w, reinterpret_cast<const unsigned short *>(a->pol_wb16), states, states_b, ld_states, mb,
                         a->head == RLPPO_HEAD_GAUSSIAN, pact, pactb, pbits, phave, pf32);
    } else {
        const float *x0 = fused_gather ? a->states : states;
        const int64_t ld0 = fused_gather ? a->ld_states : ld_states;
        const unsigned *rt = fused_gather ? rowtab : nullptr;
        rc = forward(side, val, val_w, x0, ld0, mb, 0, vact, 0, vbits, vhave, rt, src_rows, &v_folded, val_x3, true);
        if (rc) return rc;
        rc = forward(st, pol, pol_w, x0, ld0, mb, a->head == RLPPO_HEAD_GAUSSIAN, pact, 0, pbits, phave, rt, src_rows, nullptr, pol_x3);
    }
    if (rc) return rc;
    // loss epilogue: outputs -> output gradients in place, report statistics accumulated on device.  The value loss only
    // needs the critic's output and the policy loss only the policy's, so each chain runs its own loss kernel and the chains
    // do not meet until the end of the minibatch.
    LossCfg cfg;
    cfg.clip = a->clip_range;
    cfg.clip_lo = (float)(1.0 - (double)a->clip_range);
    cfg.clip_hi = (float)(1.0 + (double)a->clip_range);
    cfg.ent_coef = a->ent_coef;
    cfg.mb_ratio = a->mb_ratio;
    cfg.inv_mb = 1.0f / (float)mb;
    cfg.var_m = a->var_m;
    cfg.var_b = a->var_b;
    cfg.ring_base = ring_base;
    cfg.ring_cap = ring_cap;
    float *pout = pact[pol.n_layers - 1], *vout = vact[val.n_layers - 1];
    const int64_t ldp = pol.L[pol.n_layers - 1].pout, ldv = v_folded ? 1 : val.L[val.n_layers - 1].pout;
    rc = launch_value_loss(side, vout, ldv, nullptr, g_tgt, mb, cfg, a->stats);
    if (rc) return rc;
    float *vjoint = nullptr;  // the loss kernels' joint form (policy + value in one launch) is not used by this entry point
    if (a->head == RLPPO_HEAD_DISCRETE)
        rc = launch_discrete_loss(st, pout, ldp, n_out, vjoint, ldv, nullptr, g_act, g_old, g_tgt, g_adv, mb, cfg, a->stats);
    else if (a->head == RLPPO_HEAD_GAUSSIAN)
        rc = launch_gaussian_loss(st, pout, ldp, n_out / 2, vjoint, ldv, nullptr, g_act, g_old, g_tgt, g_adv, mb, cfg, a->stats);
    else if (a->head == RLPPO_HEAD_MULTIDISCRETE)
        rc = launch_multidiscrete_loss(st, pout, ldp, vjoint, ldv, nullptr, g_act, g_old, g_tgt, g_adv, mb, cfg, a->stats);
    else {
        set_error("ppo_minibatch: unknown head %d", a->head);
        rc = RLPPO_ERR_ARG;
    }
    if (rc) return rc;

    if (b16) {
        rc = backward_b16(side, val, val_w, reinterpret_cast<const unsigned short *>(a->val_wb16), states, states_b, ld_states, mb, vact,
                          vactb, vdx, vdxb, a->val_grad, val_tn_ws, tn_ws_floats(val, mb, prec), vbits, vhave, vf32);
        if (rc) return rc;
        rc = backward_b16(st, pol, pol_w, reinterpret_cast<const unsigned short *>(a->pol_wb16), states, states_b, ld_states, mb, pact,
                          pactb, pdx, pdxb, a->pol_grad, pol_tn_ws, tn_ws_floats(pol, mb, prec), pbits, phave, pf32);
    } else {
        ChainCtx cp, cv;
        if (fused_gather) {
            cp.rowtab = cv.rowtab = rowtab;
            cp.src = cv.src = a->states;
            cp.ld_src = cv.ld_src = a->ld_states;
            cp.src_rows = cv.src_rows = src_rows;
        }
        rc = backward(side, val, val_w, states, ld_states, mb, vact, vdx, a->val_grad, val_tn_ws, vbits, vhave, cv, v_folded, val_x3, defer);
        if (rc) return rc;
        rc = backward(st, pol, pol_w, states, ld_states, mb, pact, pdx, a->pol_grad, pol_tn_ws, pbits, phave, cp, false, pol_x3, defer);
    }
    if (rc) return rc;
    if (side != st) rc = order_after(st, side, bk.ev_join[slot]);
    if (rc) return rc;
    if (defer && dws.n) {  // [r5] every GEMM-shaped weight gradient of the pass: one launch + one reduction, after the chains have joined
        ++g_cnt_group_dw;
        rc = launch_gemm_tn_group(st, dws.p, dws.n, mb, pol_tn_ws, tn_region);
    }
    return rc;
}

int rlppo_ppo_join(void *stream) {
    SlotBank *bank = nullptr;
    int rc = slot_bank(&bank);
    if (rc) return rc;
    SlotBank &bk = *bank;
    for (int s = 1; s < RLPPO_MAX_SLOTS; ++s)
        if (bk.pending[s]) {
            rc = order_after((hipStream_t)stream, bk.main[s], bk.ev_slot[s]);
            if (rc) return rc;
            bk.pending[s] = false;
        }
    return 0;
}

int rlppo_clip_adam(void *stream, float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                    double max_norm, double lr, double beta1, double beta2, double eps, int64_t step, double *gnorm2) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && params && grads && exp_avg && exp_avg_sq && gnorm2 && step >= 1, "clip_adam: bad argument");
    // torch/optim/adam.py (_single_tensor_adam): python-double scalars, cast to fp32 where they meet a tensor
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    return launch_clip_adam((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, (float)max_norm, step_size, bc2_sqrt,
                            (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, gnorm2);
}

int rlppo_clip_adam_pack2(void *stream, const rlppo_opt_net *a, const rlppo_opt_net *b, void *sync_ws) {
    RLPPO_CHECK_ARG(a && b, "clip_adam_pack2: null descriptor");
    const rlppo_opt_net *d[2] = {a, b};
    NetLayout nets[2];
    float *p[2], *g[2], *m[2], *v[2], *packed[2];
    double *gn[2];
    int64_t n[2];
    float max_norm[2], step_size[2], bc2_sqrt[2], omb1[2], beta2[2], omb2[2], eps[2];
    for (int k = 0; k < 2; ++k) {
        int rc = make_layout(d[k]->dims, d[k]->n_layers, &nets[k]);
        if (rc) return rc;
        RLPPO_CHECK_ARG(d[k]->params && d[k]->grads && d[k]->exp_avg && d[k]->exp_avg_sq && d[k]->packed && d[k]->gnorm2 &&
                            d[k]->step >= 1, "clip_adam_pack2: bad argument (net %d)", k);
        // torch/optim/adam.py (_single_tensor_adam): python-double scalars, cast to fp32 where they meet a tensor
        const double bc1 = 1.0 - pow(d[k]->beta1, (double)d[k]->step);
        const double bc2 = 1.0 - pow(d[k]->beta2, (double)d[k]->step);
        p[k] = d[k]->params; g[k] = d[k]->grads; m[k] = d[k]->exp_avg; v[k] = d[k]->exp_avg_sq; packed[k] = d[k]->packed;
        gn[k] = d[k]->gnorm2;
        const LayerLayout &last = nets[k].L[nets[k].n_layers - 1];
        n[k] = last.off_flat_b + last.out;  // = rlppo_flat_floats
        max_norm[k] = (float)d[k]->max_norm; step_size[k] = (float)(d[k]->lr / bc1); bc2_sqrt[k] = (float)sqrt(bc2);
        omb1[k] = (float)(1.0 - d[k]->beta1); beta2[k] = (float)d[k]->beta2; omb2[k] = (float)(1.0 - d[k]->beta2);
        eps[k] = (float)d[k]->eps;
    }
    return launch_clip_adam_pack2((hipStream_t)stream, nets, p, g, m, v, packed, gn, n, max_norm, step_size, bc2_sqrt, omb1, beta2,
                                  omb2, eps, sync_ws);
}

int rlppo_learn_report(void *stream, const rlppo_report_args *a) {
    RLPPO_CHECK_ARG(a != nullptr, "learn_report: null args");
    RLPPO_CHECK_ARG(a->n_pol >= 0 && a->n_val >= 0 && (a->n_pol == 0 || (a->pol_before && a->pol_now)) && (a->n_val == 0 || (a->val_before && a->val_now)),
                    "learn_report: parameter vectors missing");
    RLPPO_CHECK_ARG(a->stats && a->out && a->done_word && a->ws, "learn_report: null pointer (stats / out / done_word / ws)");
    RLPPO_CHECK_ARG((reinterpret_cast<uintptr_t>(a->ws) & 15) == 0, "learn_report: ws must be 16-byte aligned");
    return launch_learn_report((hipStream_t)stream, *a);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ diagnostics
extern "C" {
int rlppo_gather_rows(void *stream, const float *src, int64_t ld_src, const int64_t *idx, float *dst, int32_t width, int64_t n) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && src && idx && dst, "gather_rows: null pointer");
    return launch_gather_rows((hipStream_t)stream, src, ld_src, idx, dst, width, n);
}
int rlppo_welford_increment(void *stream, const float *samples, int64_t ld, int64_t n, int32_t d, void *mean, void *m2,
                            int64_t count, int32_t state_is_f64) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0 && d > 0 && ld >= d && count >= 0 && samples && mean && m2, "welford_increment: bad argument");
    return launch_welford((hipStream_t)stream, samples, ld, n, d, mean, m2, (long long)count, state_is_f64 != 0);
}
int rlppo_welford_merge(void *stream, int32_t d, void *mean, void *m2, int64_t count, const float *other_mean,
                        const float *other_m2, int64_t other_count, int32_t state_is_f64) {
    if (other_count == 0) return 0;
    RLPPO_CHECK_ARG(d > 0 && count >= 0 && other_count > 0 && mean && m2 && other_mean && other_m2, "welford_merge: bad argument");
    return launch_welford_merge((hipStream_t)stream, d, mean, m2, (long long)count, other_mean, other_m2, (long long)other_count,
                                state_is_f64 != 0);
}
int64_t rlppo_wb16_elems(const int32_t *dims, int32_t n_layers) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return -1;
    int64_t n = 0;
    for (int l = 0; l < net.n_layers; ++l) n += (int64_t)net.L[l].pout * net.L[l].pin;
    return 2 * n;  // the W blocks, then the W^T blocks
}
int rlppo_net_pack_bf16(void *stream, const int32_t *dims, int32_t n_layers, const float *flat, float *packed_r, void *wb16) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    RLPPO_CHECK_ARG(flat && packed_r && wb16, "net_pack_bf16: null pointer");
    return launch_pack_bf16((hipStream_t)stream, net, flat, packed_r, reinterpret_cast<unsigned short *>(wb16));
}
int64_t rlppo_x3_elems(const int32_t *dims, int32_t n_layers) {
    NetLayout net;
    if (make_layout(dims, n_layers, &net)) return -1;
    X3Layout xl;
    x3_layout(net, &xl);
    return xl.total;
}
int rlppo_net_pack_x3(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, void *planes) {
    NetLayout net;
    int rc = make_layout(dims, n_layers, &net);
    if (rc) return rc;
    X3Layout xl;
    x3_layout(net, &xl);
    if (xl.total == 0) return 0;
    RLPPO_CHECK_ARG(packed && planes, "net_pack_x3: null pointer");
    unsigned short *pl = reinterpret_cast<unsigned short *>(planes);
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        if (xl.fwd[l] >= 0) {  // W [pout][pin]: the forward's B operand
            rc = launch_pack_split((hipStream_t)stream, packed + L.off_w, L.pin, L.pout, L.pin, pl + xl.fwd[l]);
            if (rc) return rc;
        }
        if (xl.dx[l] >= 0) {  // W^T [pin][pout]: dX's B operand
            rc = launch_pack_split((hipStream_t)stream, packed + L.off_wt, L.pout, L.pin, L.pout, pl + xl.dx[l]);
            if (rc) return rc;
        }
    }
    return 0;
}
// single-kernel entry points of the split-bf16 products (tests, bench.py)
int rlppo_dbg_pack_x3(void *stream, const float *S, int64_t ld, int32_t R, int32_t C, void *planes) {
    return launch_pack_split((hipStream_t)stream, S, ld, R, C, reinterpret_cast<unsigned short *>(planes));
}
int rlppo_dbg_gemm_nt_x3(void *stream, const float *A, int64_t lda, const void *planes, const float *bias, float *C, int64_t ldc, int64_t M,
                         int32_t N, int32_t K, int32_t mode, void *bits) {
    return launch_gemm_nt_split((hipStream_t)stream, A, lda, reinterpret_cast<const unsigned short *>(planes), bias, C, ldc, M, N, K, mode,
                                reinterpret_cast<unsigned long long *>(bits));
}
static int64_t g_selection_epoch = 0;  // bumped by every call that changes which kernels later launches select
int rlppo_set_update_precision(int32_t mode) {
    RLPPO_CHECK_ARG(mode >= 0 && mode <= 2, "set_update_precision: mode %d (0 = fp32, 1 = bf16 mixed precision, 2 = split-bf16 products)", mode);
    g_update_bf16 = mode;
    ++g_selection_epoch;
    return 0;
}
int rlppo_get_update_precision(void) { return g_update_bf16; }
int rlppo_set_inference_precision(int32_t mode) {
    RLPPO_CHECK_ARG(mode == 0 || mode == 1, "set_inference_precision: mode %d (0 = fp32, 1 = bf16 operands)", mode);
    set_infer_bf16(mode);
    ++g_selection_epoch;
    return 0;
}
int rlppo_dbg_set(int32_t key, int32_t value) {
    ++g_selection_epoch;
    switch (key) {
        case 1: set_gae_algo(value); return 0;
        case 4: g_two_streams = value; return 0;
        case 21: set_gae_spin_limit(value); return 0;
        case 22: set_gae_oversubscribe(value); return 0;
        case 34: set_fused_spin_limit(value); return 0;
        case 35: set_fused_test_hold(value); return 0;
        case 36: set_split_persistent(value); return 0;
        case 23: set_b16_wide_tiles(value); return 0;
        case 24: set_exp_fast_transform(value); return 0;
        case 26: g_fused_gather = value; return 0;
        case 27: g_fused_act = value; return 0;
        case 29: g_paired = value; return 0;
        case 31: g_head_order = value; return 0;
        case 32: g_fold_vhead = value; return 0;
        case 33: set_pair_interleave(value); return 0;
        case 37: g_group_dw = value; return 0;
        case 38: set_tn_group_budget(value); return 0;
        case 40: set_nt_skinny(value); return 0;
        case 41: g_window_mode = value; return 0;
        default: break;
    }
    set_error("dbg_set: unknown key %d", key);
    return RLPPO_ERR_ARG;
}
int64_t rlppo_selection_epoch(void) { return g_selection_epoch; }
const int64_t *rlppo_selection_epoch_ptr(void) { return &g_selection_epoch; }
int64_t rlppo_dbg_counter(int32_t key) {
    switch (key) {
        case 0: return g_cnt_fused_act;
        case 1: return g_cnt_act_chain;
        case 2: return g_cnt_pass;
        case 3: return g_cnt_paired_pass;
        case 4: return g_cnt_gather_fused_pass;
        case 5: return g_cnt_group_dw;
        default: return -1;
    }
}
size_t rlppo_dbg_gemm_nt_bits_bytes(int64_t M, int32_t N) { return nt_bits_floats(M, N) * sizeof(float); }
int rlppo_dbg_gemm_nt_bits(void *stream, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                           int64_t ldc, int64_t M, int32_t N, int32_t K, int32_t epilogue, void *bits) {
    const int rc = launch_gemm_nt_bits((hipStream_t)stream, A, lda, B, ldb, bias, C, ldc, M, N, K, epilogue,
                                       reinterpret_cast<unsigned long long *>(bits));
    if (rc == -1) {
        set_error("dbg_gemm_nt_bits: the bitmask form does not apply to this call");
        return RLPPO_ERR_ARG;
    }
    return rc;
}
int rlppo_dbg_gemm_nt(void *stream, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                      const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                      int32_t epilogue) {
    return launch_gemm_nt((hipStream_t)stream, A, lda, B, ldb, bias, mask_src, ld_mask, C, ldc, M, N, K, epilogue);
}
int rlppo_dbg_gemm_nt_b16(void *stream, const void *A, int64_t lda, const void *W, int64_t ldw, const float *bias, float *C,
                          int64_t ldc, void *Cb, int64_t ldcb, int64_t M, int32_t N, int32_t K, int32_t epilogue, int32_t hidden,
                          void *bits) {
    return launch_gemm_nt_b16((hipStream_t)stream, reinterpret_cast<const unsigned short *>(A), lda,
                              reinterpret_cast<const unsigned short *>(W), ldw, bias, C, ldc, reinterpret_cast<unsigned short *>(Cb),
                              ldcb, M, N, K, epilogue, hidden, reinterpret_cast<unsigned long long *>(bits));
}
int rlppo_dbg_gemm_tn_b16(void *stream, const void *dY, int64_t ldy, const void *X, int64_t ldx, float *dW, float *db, int32_t pout,
                          int32_t pin, int32_t out, int32_t in, int64_t M, void *ws, size_t ws_bytes) {
    return launch_gemm_tn_b16((hipStream_t)stream, reinterpret_cast<const unsigned short *>(dY), ldy,
                              reinterpret_cast<const unsigned short *>(X), ldx, dW, db, pout, pin, out, in, M, (float *)ws,
                              ws_bytes / sizeof(float));
}
size_t rlppo_dbg_thin_head_workspace_bytes(int32_t out, int32_t kp, int64_t M) { return thin_dw_ws_floats(out, kp, M) * sizeof(float); }
int rlppo_dbg_thin_head_b16(void *stream, const float *dY, int64_t ldy, int32_t out, const float *W, int64_t ldw, const void *bits,
                            const void *hb, int64_t ldh, void *dxb, float *dW, float *db, int32_t in, int32_t kp, int64_t M, void *ws,
                            size_t ws_bytes) {
    int rc = launch_thin_dx_b16((hipStream_t)stream, dY, ldy, out, W, ldw, reinterpret_cast<const unsigned long long *>(bits),
                                reinterpret_cast<unsigned short *>(dxb), kp, kp, M);
    if (rc) return rc == -1 ? RLPPO_ERR_ARG : rc;
    rc = launch_thin_dw_b16((hipStream_t)stream, dY, ldy, reinterpret_cast<const unsigned short *>(hb), ldh, dW, db, out, in, kp, M,
                            (float *)ws, ws_bytes / sizeof(float));
    return rc == -1 ? RLPPO_ERR_ARG : rc;
}
static int tn_products(const rlppo_tn_product *p, int32_t n, TnProduct *q) {
    RLPPO_CHECK_ARG(p && n > 0 && n <= 2 * RLPPO_MAX_LAYERS, "gemm_tn_group: %d products (1..%d)", n, 2 * RLPPO_MAX_LAYERS);
    for (int i = 0; i < n; ++i) {
        q[i].dY = p[i].dY; q[i].ldy = p[i].ldy; q[i].ny_valid = p[i].ny_valid; q[i].X = p[i].X; q[i].ldx = p[i].ldx;
        q[i].kx_valid = p[i].kx_valid; q[i].dW = p[i].dW; q[i].db = p[i].db; q[i].out = p[i].out; q[i].in = p[i].in;
        q[i].rowtab = p[i].rowtab; q[i].src_rows = p[i].src_rows;
    }
    return 0;
}
size_t rlppo_dbg_gemm_tn_group_workspace_bytes(const rlppo_tn_product *p, int32_t n, int64_t M) {
    if (!p || n <= 0 || n > 2 * RLPPO_MAX_LAYERS) return 0;
    int outs[2 * RLPPO_MAX_LAYERS], ins[2 * RLPPO_MAX_LAYERS];
    for (int i = 0; i < n; ++i) {
        outs[i] = p[i].out;
        ins[i] = p[i].in;
    }
    return tn_group_floats(outs, ins, n, M) * sizeof(float);
}
int rlppo_dbg_gemm_tn_group(void *stream, const rlppo_tn_product *p, int32_t n, int64_t M, void *ws, size_t ws_bytes) {
    TnProduct q[2 * RLPPO_MAX_LAYERS];
    const int rc = tn_products(p, n, q);
    if (rc) return rc;
    return launch_gemm_tn_group((hipStream_t)stream, q, n, M, (float *)ws, ws_bytes / sizeof(float));
}
size_t rlppo_dbg_gemm_tn_workspace_bytes(int32_t out, int32_t in, int64_t M) { return tn_partial_floats(out, in, M) * sizeof(float); }
int rlppo_dbg_gemm_tn(void *stream, const float *dY, int64_t ldy, int32_t ny_valid, const float *X, int64_t ldx,
                      int32_t kx_valid, float *dW, float *db, int32_t out, int32_t in, int64_t M, void *ws, size_t ws_bytes) {
    return launch_gemm_tn((hipStream_t)stream, dY, ldy, ny_valid, X, ldx, kx_valid, dW, db, out, in, M, (float *)ws,
                          ws_bytes / sizeof(float));
}
}
