// optim.hip -- parameter-side kernels: packing the flat arena into tile-padded W / W^T / b, observation row
// padding (+ the reference's scalar standardisation), gradient-norm clip + Adam on the flat arena.
#include <atomic>

#include "common.hpp"

namespace rlppo {

// ------------------------------------------------------------------------------------------------ pack
struct PackJob {
    int in, out, pin, pout;
    int64_t off_w, off_wt, off_b, off_flat_w, off_flat_b;
    int64_t first;  // first global work item of this layer
};
struct PackJobs {
    int n;
    PackJob j[RLPPO_MAX_LAYERS];
    int64_t total;
};

// one work item per element of the padded W (pout*pin); the same thread also writes W^T and (row 0 items) b
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ flat, float *__restrict__ packed, PackJobs jobs) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < jobs.total; g += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < jobs.n && g >= jobs.j[l + 1].first) ++l;
        const PackJob &J = jobs.j[l];
        const int64_t e = g - J.first;
        const int o = (int)(e / J.pin), i = (int)(e % J.pin);
        const float w = (o < J.out && i < J.in) ? flat[J.off_flat_w + (int64_t)o * J.in + i] : 0.f;
        packed[J.off_w + (int64_t)o * J.pin + i] = w;
        packed[J.off_wt + (int64_t)i * J.pout + o] = w;
        if (i == 0) packed[J.off_b + o] = o < J.out ? flat[J.off_flat_b + o] : 0.f;
    }
}

int launch_pack(hipStream_t st, const NetLayout &net, const float *flat, float *packed) {
    PackJobs jobs;
    jobs.n = net.n_layers;
    int64_t tot = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        jobs.j[l] = PackJob{L.in, L.out, L.pin, L.pout, L.off_w, L.off_wt, L.off_b, L.off_flat_w, L.off_flat_b, tot};
        tot += (int64_t)L.pin * L.pout;
    }
    jobs.total = tot;
    const int blocks = (int)(cdiv(tot, 256) < 2048 ? cdiv(tot, 256) : 2048);
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, flat, packed, jobs);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// bf16 update precision: the fp32 master copy rounded to bf16 (round-to-nearest-even), once per optimiser step:
//   packed_r -- the packed image (W | W^T | b per layer) holding the ROUNDED weights as fp32 (the fp32 dX product and the
//               critic's matrix-vector head then multiply exactly what the bf16 forward multiplied); biases stay fp32;
//   wb16     -- the W[Pout][Pin] blocks alone as bf16, layer after layer (the B operand of the forward gemm_nt_b16_kernel),
//               followed by the W^T[Pin][Pout] blocks in the same order (the B operand of its dX form).
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float *__restrict__ flat, float *__restrict__ packed_r,
                                                        unsigned short *__restrict__ wb16, PackJobs jobs) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < jobs.total; g += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < jobs.n && g >= jobs.j[l + 1].first) ++l;
        const PackJob &J = jobs.j[l];
        const int64_t e = g - J.first;
        const int o = (int)(e / J.pin), i = (int)(e % J.pin);
        const float w = (o < J.out && i < J.in) ? flat[J.off_flat_w + (int64_t)o * J.in + i] : 0.f;
        const unsigned short h = bf16_bits(w);
        const float wr = __uint_as_float((unsigned)h << 16);
        packed_r[J.off_w + (int64_t)o * J.pin + i] = wr;
        packed_r[J.off_wt + (int64_t)i * J.pout + o] = wr;
        wb16[J.first + e] = h;  // J.first = sum of the previous layers' Pout * Pin
        wb16[jobs.total + J.first + (int64_t)i * J.pout + o] = h;
        if (i == 0) packed_r[J.off_b + o] = o < J.out ? flat[J.off_flat_b + o] : 0.f;
    }
}

int launch_pack_bf16(hipStream_t st, const NetLayout &net, const float *flat, float *packed_r, unsigned short *wb16) {
    PackJobs jobs;
    jobs.n = net.n_layers;
    int64_t tot = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        jobs.j[l] = PackJob{L.in, L.out, L.pin, L.pout, L.off_w, L.off_wt, L.off_b, L.off_flat_w, L.off_flat_b, tot};
        tot += (int64_t)L.pin * L.pout;
    }
    jobs.total = tot;
    const int blocks = (int)(cdiv(tot, 256) < 2048 ? cdiv(tot, 256) : 2048);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(blocks), dim3(256), 0, st, flat, packed_r, wb16, jobs);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// x[r][0..width) -> rounded to bf16 in place (as fp32) + the bf16 copy xb[r][0..width): the hand-over of a layer that took the
// fp32 kernels in the bf16 update precision (shapes gemm_nt_b16_kernel does not cover).  4 elements per thread.
__global__ __launch_bounds__(256) void round_rows_kernel(float *__restrict__ x, unsigned short *__restrict__ xb, int64_t n4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<f32x4 *>(x)[g];
        u16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = v[e];
            h[e] = bf16_bits(f);
            v[e] = __uint_as_float((unsigned)h[e] << 16);
        }
        reinterpret_cast<f32x4 *>(x)[g] = v;
        reinterpret_cast<u16x4 *>(xb)[g] = h;
    }
}

int launch_round_rows(hipStream_t st, float *x, unsigned short *xb, int64_t n_elems) {
    if (n_elems <= 0) return 0;
    RLPPO_CHECK_ARG(n_elems % 4 == 0, "round_rows: element count %ld is not a multiple of 4", (long)n_elems);
    const int64_t n4 = n_elems / 4;
    const int blocks = (int)(cdiv(n4, 256) < 8192 ? cdiv(n4, 256) : 8192);
    hipLaunchKernelGGL(round_rows_kernel, dim3(blocks), dim3(256), 0, st, x, xb, n4);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// xb (bf16) -> x (the same values as fp32): the hand-over from a bf16-in-memory kernel to a layer that takes the fp32 kernels.
__global__ __launch_bounds__(256) void expand_rows_kernel(const unsigned short *__restrict__ xb, float *__restrict__ x, int64_t n4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (int64_t)gridDim.x * blockDim.x) {
        const u16x4 h = reinterpret_cast<const u16x4 *>(xb)[g];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float((unsigned)h[e] << 16);
        reinterpret_cast<f32x4 *>(x)[g] = v;
    }
}

int launch_expand_rows(hipStream_t st, const unsigned short *xb, float *x, int64_t n_elems) {
    if (n_elems <= 0) return 0;
    RLPPO_CHECK_ARG(n_elems % 4 == 0, "expand_rows: element count %ld is not a multiple of 4", (long)n_elems);
    const int64_t n4 = n_elems / 4;
    const int blocks = (int)(cdiv(n4, 256) < 8192 ? cdiv(n4, 256) : 8192);
    hipLaunchKernelGGL(expand_rows_kernel, dim3(blocks), dim3(256), 0, st, xb, x, n4);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// action indices as floats (the experience buffer's encoding, experience_buffer.py:72) -- the launch-by-launch form of
// rlppo_discrete_step; the fused kernel writes them itself
__global__ __launch_bounds__(256) void i64_to_f32_kernel(const int64_t *__restrict__ src, float *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
// [r5] rlppo_act_opts.done_words for the calls whose last kernel does not write them itself: stream order puts this launch behind
// every output of the call; the words are stored with release semantics at system scope (visible to a polling host)
__global__ void signal_words_kernel(unsigned *__restrict__ words, int count, unsigned value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    for (int i = threadIdx.x; i < count; i += blockDim.x) __hip_atomic_store(words + i, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_signal_words(hipStream_t st, unsigned *words, int count, unsigned value) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(signal_words_kernel, dim3(1), dim3(64), 0, st, words, count, value);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
int launch_i64_to_f32(hipStream_t st, const int64_t *src, float *dst, int64_t n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(i64_to_f32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, src, dst, n);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------------------------- pad rows
template <typename T>
__global__ __launch_bounds__(256) void pad_rows_kernel(const T *__restrict__ src, int64_t n, int64_t d, int64_t ld_src,
                                                        float *__restrict__ dst, int64_t ld_dst, int standardize,
                                                        float mean0, float std0) {
    const int64_t total = n * ld_dst;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = g / ld_dst, c = g % ld_dst;
        float v = 0.f;
        if (c < d) {
            v = (float)src[r * ld_src + c];
            // np.clip((obs - mean) / std, -5, 5) with float32 scalars (batched_agent_manager.py:313-315)
            if (standardize) v = fminf(fmaxf((v - mean0) / std0, -5.f), 5.f);
        }
        dst[g] = v;
    }
}

// The same copy with PER-FEATURE statistics: dst = clip((src - mean[c]) / std[c], -5, 5).  The reference standardises every
// feature with the scalars of feature 0 (quirk Q5, above); this is the form its obs_stats were meant for, behind a flag.
template <typename T>
__global__ __launch_bounds__(256) void pad_rows_vec_kernel(const T *__restrict__ src, int64_t n, int64_t d, int64_t ld_src,
                                                            float *__restrict__ dst, int64_t ld_dst,
                                                            const float *__restrict__ mean, const float *__restrict__ stdv) {
    const int64_t total = n * ld_dst;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = g / ld_dst, c = g % ld_dst;
        float v = 0.f;
        if (c < d) v = fminf(fmaxf(((float)src[r * ld_src + c] - mean[c]) / stdv[c], -5.f), 5.f);
        dst[g] = v;
    }
}

// Minibatch gather (experience_buffer.py:82-87): dst[r][0..width) = src[idx[r]][0..width), 16 bytes per thread.  One pass
// per minibatch, shared by the policy and the critic: the four first-layer GEMMs (two forwards, two dW) then read
// contiguous rows through the LDS-DMA kernels instead of each chasing the index vector.
// [r5] `meta` (optional): the per-row scalars of gather_meta_kernel ride along -- the thread of a row's chunk 0 gathers them -- so a
// pass that gathers its rows (every pass below 262,144 rows) starts with ONE launch instead of two.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, int64_t ld_src,
                                                          const int64_t *__restrict__ idx, float *__restrict__ dst,
                                                          int chunks_per_row, int64_t n, int64_t ring_base, int64_t ring_cap, GatherMeta meta) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / chunks_per_row;
    const int c = (int)(t - row * chunks_per_row);
    if (row >= n) return;
    const int64_t srow = ring_row(idx[row], ring_base, ring_cap);
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + srow * ld_src) + c);
    reinterpret_cast<f32x4 *>(dst)[t] = v;
    if (c == 0 && meta.g_old) {
        meta.g_old[row] = meta.old_logp[srow];
        meta.g_adv[row] = meta.adv[srow];
        meta.g_tgt[row] = meta.targets[srow];
        for (int k = 0; k < meta.act_dim; ++k) meta.g_act[row * meta.act_dim + k] = meta.actions[srow * meta.act_dim + k];
        if (meta.zero_n) meta.zero_n[row] = 0.f;
    }
}

// The same gather for the bf16 update precision: the gathered observation rows are rounded to bf16 on the way -- written as
// fp32 (X operand of the first layer's fp32 weight gradient) and as bf16 (A operand of the first layer's bf16 forward).
__global__ __launch_bounds__(256) void gather_rows_round_kernel(const float *__restrict__ src, int64_t ld_src,
                                                                const int64_t *__restrict__ idx, float *__restrict__ dst,
                                                                unsigned short *__restrict__ dstb, int chunks_per_row, int64_t n,
                                                                int64_t ring_base, int64_t ring_cap) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / chunks_per_row;
    const int c = (int)(t - row * chunks_per_row);
    if (row >= n) return;
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + ring_row(idx[row], ring_base, ring_cap) * ld_src) + c);
    u16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float f = v[e];
        h[e] = bf16_bits(f);
        v[e] = __uint_as_float((unsigned)h[e] << 16);
    }
    if (dst) reinterpret_cast<f32x4 *>(dst)[t] = v;  // the fp32 form only when an fp32 kernel will read the rows
    reinterpret_cast<u16x4 *>(dstb)[t] = h;
}

int launch_gather_rows_round(hipStream_t st, const float *src, int64_t ld_src, const int64_t *idx, float *dst, unsigned short *dstb,
                             int width, int64_t n, int64_t ring_base, int64_t ring_cap) {
    if (n <= 0) return 0;
    RLPPO_CHECK_ARG(width > 0 && width % 4 == 0 && ld_src >= width && ld_src % 4 == 0, "gather_rows: width=%d ld=%ld", width,
                    (long)ld_src);
    const int cpr = width / 4;
    hipLaunchKernelGGL(gather_rows_round_kernel, dim3((unsigned)cdiv(n * cpr, 256)), dim3(256), 0, st, src, ld_src, idx, dst, dstb,
                       cpr, n, ring_base, ring_cap);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// The per-row scalars of a minibatch (experience_buffer.py:82-87 gathers them with the states): actions[act_dim], old log-prob,
// advantage, value target of row idx[r] -> four contiguous arrays.  One thread per row, so the dependent idx -> row loads of
// the whole minibatch are in flight together; the loss kernels then stream contiguous data instead of chasing the index
// inside their row loop (two dependent HBM round trips per row with 16 rows per lane group: 150 us per 524,288-row pass).
__global__ __launch_bounds__(256) void gather_meta_kernel(const int64_t *__restrict__ idx, const float *__restrict__ actions,
                                                          int act_dim, const float *__restrict__ old_logp,
                                                          const float *__restrict__ adv, const float *__restrict__ targets,
                                                          float *__restrict__ g_act, float *__restrict__ g_old,
                                                          float *__restrict__ g_adv, float *__restrict__ g_tgt, int64_t n,
                                                          int64_t ring_base, int64_t ring_cap, unsigned *__restrict__ rowtab,
                                                          float *__restrict__ zero_n) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int64_t src = ring_row(idx[r], ring_base, ring_cap);
    if (rowtab) rowtab[r] = (unsigned)src;  // [r3] the row table of the fused state gather (gemm.hip, GATHER)
    if (zero_n) zero_n[r] = 0.f;            // [r5] the critic's folded value head accumulates into zeros: no fill launch of its own (23 us at 65,536 rows)
    g_old[r] = old_logp[src];
    g_adv[r] = adv[src];
    g_tgt[r] = targets[src];
    for (int k = 0; k < act_dim; ++k) g_act[r * act_dim + k] = actions[src * act_dim + k];
}

int launch_gather_meta(hipStream_t st, const int64_t *idx, const float *actions, int act_dim, const float *old_logp,
                       const float *adv, const float *targets, float *g_act, float *g_old, float *g_adv, float *g_tgt, int64_t n,
                       int64_t ring_base, int64_t ring_cap, unsigned *rowtab, float *zero_n) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_meta_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, idx, actions, act_dim, old_logp, adv,
                       targets, g_act, g_old, g_adv, g_tgt, n, ring_base, ring_cap, rowtab, zero_n);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_gather_rows(hipStream_t st, const float *src, int64_t ld_src, const int64_t *idx, float *dst, int width,
                       int64_t n, int64_t ring_base, int64_t ring_cap, const GatherMeta *meta) {
    if (n <= 0) return 0;
    RLPPO_CHECK_ARG(width > 0 && width % 4 == 0 && ld_src >= width && ld_src % 4 == 0, "gather_rows: width=%d ld=%ld", width,
                    (long)ld_src);
    const int cpr = width / 4;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv(n * cpr, 256)), dim3(256), 0, st, src, ld_src, idx, dst, cpr, n, ring_base,
                       ring_cap, meta ? *meta : GatherMeta{});
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// WelfordRunningStat.increment (running_stats.py:28-46) for n samples of d features: the reference updates sample by
// sample in float32, and each feature's recurrence is independent of the others -- so one thread per feature walking the
// samples in order reproduces it BIT FOR BIT (IEEE division, no contraction), while the d threads read the sample rows
// coalesced.  4096 samples take ~0.1 ms on the stream instead of ~10 ms of Python on the host.
// T = the dtype of the running state: float32 as constructed, float64 after WelfordRunningStat.from_json (np.asarray of
// Python floats; the reference then updates in float64: float32 sample - float64 mean -> float64).
template <typename T>
__global__ __launch_bounds__(128) void welford_kernel(const float *__restrict__ x, int64_t ld, int64_t n, int d,
                                                      T *__restrict__ mean, T *__restrict__ m2, long long count0) {
    const int f = blockIdx.x * 128 + threadIdx.x;
    if (f >= d) return;
    T mu = mean[f], v = m2[f];
    // the recurrence is serial in mu / v, the loads are not: 16 samples are requested together ([r3]: with one dependent load
    // per iteration 4096 samples took 0.86 ms, a third of a 4096-agent collect; the chain of IEEE divisions alone is ~0.1 ms)
    constexpr int PF = 16;
    int64_t i = 0;
    for (; i + PF <= n; i += PF) {
        float xs[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) xs[u] = x[(i + u) * ld + f];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const long long prev = count0 + i + u;
            const T cnt = (T)(prev + 1);
            const T delta = (T)xs[u] - mu;            // delta   = sample - running_mean
            const T dn = delta / cnt;                 // delta_n = delta / count
            mu += dn;                                 // running_mean += delta_n
            v += (delta * dn) * (T)prev;              // running_variance += delta * delta_n * (count - 1)
        }
    }
    for (; i < n; ++i) {
        const long long prev = count0 + i;
        const T cnt = (T)(prev + 1);
        const T delta = (T)x[i * ld + f] - mu;
        const T dn = delta / cnt;
        mu += dn;
        v += (delta * dn) * (T)prev;
    }
    mean[f] = mu;
    m2[f] = v;
}

int launch_welford(hipStream_t st, const float *x, int64_t ld, int64_t n, int d, void *mean, void *m2, long long count0,
                   int state_f64) {
    if (n <= 0 || d <= 0) return 0;
    if (state_f64)
        hipLaunchKernelGGL(welford_kernel<double>, dim3((unsigned)cdiv(d, 128)), dim3(128), 0, st, x, ld, n, d, (double *)mean,
                           (double *)m2, count0);
    else
        hipLaunchKernelGGL(welford_kernel<float>, dim3((unsigned)cdiv(d, 128)), dim3(128), 0, st, x, ld, n, d, (float *)mean,
                           (float *)m2, count0);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// WelfordRunningStat.increment_from_serialized_other (running_stats.py:71-98): Chan's parallel combination of two running
// statistics, elementwise over the d features, in the state's dtype and in the reference's operation order:
//   delta = other_mean - mean;  mean' = (count*mean + other_count*other_mean) / total
//   m2'   = m2 + other_m2 + ((delta*delta)*count)*other_count / total
template <typename T>
__global__ __launch_bounds__(128) void welford_merge_kernel(int d, T *__restrict__ mean, T *__restrict__ m2, long long count,
                                                            const float *__restrict__ omean, const float *__restrict__ om2,
                                                            long long ocount) {
    const int f = blockIdx.x * 128 + threadIdx.x;
    if (f >= d) return;
    const T c = (T)count, oc = (T)ocount, tot = (T)(count + ocount);
    const T mu = mean[f], om = (T)omean[f];
    const T delta = om - mu;
    const T dsq = delta * delta;
    // `other_count * other_mean` is int x float32-array = a float32 product whatever the state's dtype (the reference casts
    // the serialised list to float32, running_stats.py:79-80); only then does it meet the (possibly float64) state
    const T oterm = (T)((float)ocount * omean[f]);
    mean[f] = (c * mu + oterm) / tot;
    m2[f] = (m2[f] + (T)om2[f]) + ((dsq * c) * oc) / tot;
}

int launch_welford_merge(hipStream_t st, int d, void *mean, void *m2, long long count, const float *omean, const float *om2,
                         long long ocount, int state_f64) {
    if (d <= 0 || ocount == 0) return 0;
    if (state_f64)
        hipLaunchKernelGGL(welford_merge_kernel<double>, dim3((unsigned)cdiv(d, 128)), dim3(128), 0, st, d, (double *)mean,
                           (double *)m2, count, omean, om2, ocount);
    else
        hipLaunchKernelGGL(welford_merge_kernel<float>, dim3((unsigned)cdiv(d, 128)), dim3(128), 0, st, d, (float *)mean,
                           (float *)m2, count, omean, om2, ocount);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_pad_rows(hipStream_t st, const void *src, int is_f64, int64_t n, int64_t d, int64_t ld_src, float *dst,
                    int64_t ld_dst, int standardize, float mean0, float std0) {
    if (n <= 0) return 0;
    const int64_t total = n * ld_dst;
    const int blocks = (int)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    if (is_f64)
        hipLaunchKernelGGL(pad_rows_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double *)src, n, d, ld_src, dst,
                           ld_dst, standardize, mean0, std0);
    else
        hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)src, n, d, ld_src, dst,
                           ld_dst, standardize, mean0, std0);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_pad_rows_vec(hipStream_t st, const void *src, int is_f64, int64_t n, int64_t d, int64_t ld_src, float *dst,
                        int64_t ld_dst, const float *mean, const float *stdv) {
    if (n <= 0) return 0;
    const int64_t total = n * ld_dst;
    const int blocks = (int)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    if (is_f64)
        hipLaunchKernelGGL(pad_rows_vec_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double *)src, n, d, ld_src, dst,
                           ld_dst, mean, stdv);
    else
        hipLaunchKernelGGL(pad_rows_vec_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)src, n, d, ld_src, dst,
                           ld_dst, mean, stdv);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ clip + Adam
__global__ __launch_bounds__(256) void sqnorm_kernel(const float *__restrict__ g, int64_t n, double *__restrict__ out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)g[i];
        acc += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.Adam single-tensor math (lerp / mul+addcmul / sqrt,div,add / addcdiv), with the clip coefficient
// of clip_grad_norm_ folded in: coef = min(1, max_norm / (||g|| + 1e-6)).
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, int64_t n, const double *__restrict__ gnorm2,
                                                    float max_norm, float step_size, float bc2_sqrt, float omb1,
                                                    float beta2, float omb2, float eps) {
    const float total = (float)sqrt(*gnorm2);
    float coef = max_norm / (total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        g[i] = gi;  // clip_grad_norm_ scales .grad in place
        float mi = m[i], vi = v[i];
        mi = mi + omb1 * (gi - mi);             // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2 + (omb2 * gi) * gi;       // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + (-step_size * mi) / denom;
        m[i] = mi;
        v[i] = vi;
    }
}

int launch_clip_adam(hipStream_t st, float *p, float *g, float *m, float *v, int64_t n, float max_norm, float step_size,
                     float bc2_sqrt, float omb1, float beta2, float omb2, float eps, double *gnorm2) {
    if (n <= 0) return 0;
    RLPPO_HIP(hipMemsetAsync(gnorm2, 0, sizeof(double), st));
    const int blocks = (int)(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, st, g, n, gnorm2);
    RLPPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, gnorm2, max_norm, step_size, bc2_sqrt,
                       omb1, beta2, omb2, eps);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------- one optimiser step of BOTH networks, fused
// PPOLearner.learn's tail per batch is value_optimizer.step(); policy_optimizer.step() after two clip_grad_norm_ calls
// (ppo_learner.py:187-193), followed here by the re-packing of the kernels' weight copy and, at the next batch, by
// zero_grad.  As separate launches that is 2 memsets + 2 norms + 2 Adam + 2 packs + 1 fill = 9 dependent stream operations
// of 5-10 us each: ~60 us, 4-5 % of the 1.3 ms a rank of an 8-rank job spends per optimiser step.  Here: one memset, one
// launch for both squared norms, one launch that does clip + Adam for both arenas, writes every updated parameter straight
// into its W / W^T / b slots of the packed copy (the zero padding of the packed copy is never touched) and leaves the
// gradient arena zeroed for the next batch.  Same arithmetic, element by element, as adam_kernel / pack_kernel.
struct OptNet {
    float *p, *g, *m, *v, *packed;
    double *gnorm2;
    int64_t n;
    float max_norm, step_size, bc2_sqrt, omb1, beta2, omb2, eps;
    PackJobs jobs;
};
struct OptPair {
    OptNet net[2];
};

__global__ __launch_bounds__(256) void sqnorm2_kernel(OptPair o) {
    const int k = blockIdx.y;
    const float *g = o.net[k].g;
    const int64_t n = o.net[k].n;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)g[i];
        acc += v * v;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(o.net[k].gnorm2, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adam_pack2_kernel(OptPair o) {
    const OptNet &N = o.net[blockIdx.y];
    const float total = (float)sqrt(*N.gnorm2);
    float coef = N.max_norm / (total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = N.g[i] * coef;
        float mi = N.m[i], vi = N.v[i];
        mi = mi + N.omb1 * (gi - mi);
        vi = vi * N.beta2 + (N.omb2 * gi) * gi;
        const float denom = sqrtf(vi) / N.bc2_sqrt + N.eps;
        const float pi = N.p[i] + (-N.step_size * mi) / denom;
        N.p[i] = pi;
        N.m[i] = mi;
        N.v[i] = vi;
        N.g[i] = 0.f;  // zero_grad of the next batch (the clipped gradient is not observable inside learn())
        int l = 0;     // flat order: W0[out][in], b0, W1, b1, ...
        while (l + 1 < N.jobs.n && i >= N.jobs.j[l + 1].off_flat_w) ++l;
        const PackJob &J = N.jobs.j[l];
        if (i < J.off_flat_b) {
            const int64_t e = i - J.off_flat_w;
            const int r = (int)(e / J.in), c = (int)(e % J.in);
            N.packed[J.off_w + (int64_t)r * J.pin + c] = pi;
            N.packed[J.off_wt + (int64_t)c * J.pout + r] = pi;
        } else {
            N.packed[J.off_b + (i - J.off_flat_b)] = pi;
        }
    }
}

// ---- the same update in ONE launch: squared norms -> grid barrier -> clip + Adam + re-pack + zero_grad.
// The three-operation form above costs, per optimiser step of an 8-rank job (profiles/r03_rank_share_step_gaps_before.txt): a
// 6 us fill of the two accumulators, 19 us for sqnorm2 (2048 workgroups queueing on two double-precision atomics) and 9 us for
// adam_pack2, each behind a launch boundary with the chip idle.  Here every workgroup writes its partial sum of squares into a
// slot of its own in the caller's sync block and arrives at a counter; the LAST workgroup to arrive adds the slots of both
// networks IN A FIXED ORDER (so the norm -- and with it every replica of a data-parallel job -- is bit-reproducible, which two
// atomically accumulated doubles are not), publishes the totals to the nets' gnorm2 outputs, re-arms the counter and opens the
// barrier (generation word).  While they wait the others already hold the Adam operands of their first elements in registers.
// Agent-scope atomics only on the words that cross workgroups (cdna_hip_programming.md Guideline 16); the block needs no
// per-launch memset: it is zeroed ONCE by its owner and every completed launch leaves it armed.  All workgroups must be
// co-resident: the grid is clamped per device to (occupancy x CUs - one workgroup per CU of margin) / 2 per network and to
// FUSED_MAX_BLOCKS (launch_clip_adam_pack2; below one workgroup the three-operation form runs instead).
// [r4] A wait that gives up is ALL OR NOTHING and leaves no NaN behind: opening the barrier and giving up are both compare-and-swaps
// on the generation word (g0 -> g0 + 1 | g0 -> FUSED_DEAD), so exactly one of them happens; after FUSED_DEAD every workgroup of this
// launch -- waiting, arriving late, or the last arriver itself -- skips its update: parameters, moments and gradients of BOTH
// networks stay as they were, the block's timeout word counts the event, and every later launch on that block skips at once
// until its owner has zeroed it again.  (Round 3 poisoned the step with NaN, which could not be recovered from: advisor finding.)
struct FusedSync {
    unsigned count, gen, timeouts, counted_seq, pad[12];  // 64-byte header; counted_seq: the last launch that added itself to `timeouts`
    double slot[2][512];                     // per network, per workgroup: partial sum of squares of this launch
};
static_assert(sizeof(FusedSync) <= RLPPO_OPT_SYNC_BYTES, "RLPPO_OPT_SYNC_BYTES");
constexpr int FUSED_EPT = 6;                    // elements per thread whose operands are fetched before the barrier
constexpr int FUSED_MAX_BLOCKS = 128;           // per network: 256 workgroups per launch, so that even 8 processes sharing one GPU (the
                                                // gloo tests) keep every launch's grid co-resident (2048 workgroup slots of 256 threads)
constexpr unsigned FUSED_SPIN_LIMIT = 1u << 22;  // x (s_sleep 8 + one L2 round trip) ~ seconds
constexpr unsigned FUSED_DEAD = 0xFFFFFFFFu;     // generation word of a block whose barrier was given up (sticky until re-zeroed)
static unsigned g_fused_spin_limit = FUSED_SPIN_LIMIT;  // rlppo_dbg_set(34, v): tests force the give-up path with 0
void set_fused_spin_limit(int v) { g_fused_spin_limit = v < 0 ? FUSED_SPIN_LIMIT : (unsigned)v; }
static int g_fused_test_hold = 0;  // rlppo_dbg_set(35, 1): test hook -- workgroup (0, 0) never arrives (a grid that is not co-resident)
void set_fused_test_hold(int v) { g_fused_test_hold = v; }

__device__ __forceinline__ void adam_one(const OptNet &N, int64_t i, float gi, float mi, float vi, float p0, float coef) {
    gi *= coef;
    mi = mi + N.omb1 * (gi - mi);
    vi = vi * N.beta2 + (N.omb2 * gi) * gi;
    const float denom = sqrtf(vi) / N.bc2_sqrt + N.eps;
    const float pi = p0 + (-N.step_size * mi) / denom;
    N.p[i] = pi;
    N.m[i] = mi;
    N.v[i] = vi;
    N.g[i] = 0.f;  // zero_grad of the next batch (the clipped gradient is not observable inside learn())
    int l = 0;     // flat order: W0[out][in], b0, W1, b1, ...
    while (l + 1 < N.jobs.n && i >= N.jobs.j[l + 1].off_flat_w) ++l;
    const PackJob &J = N.jobs.j[l];
    if (i < J.off_flat_b) {
        const int64_t e = i - J.off_flat_w;
        const int r = (int)(e / J.in), c = (int)(e % J.in);
        N.packed[J.off_w + (int64_t)r * J.pin + c] = pi;
        N.packed[J.off_wt + (int64_t)c * J.pout + r] = pi;
    } else {
        N.packed[J.off_b + (i - J.off_flat_b)] = pi;
    }
}

// `timeouts` counts SKIPPED OPTIMISER STEPS (the host rewinds Adam's step count by it), so a launch adds itself at most once:
// the waiter that wins the give-up and workgroup (0, 0) finding the block dead can be the same launch (a late-dispatched (0, 0)
// on a grid that was not co-resident -- exactly the give-up case); whoever swaps the launch's sequence number in first counts.
__device__ __forceinline__ void count_skipped_once(FusedSync *sy, unsigned seq) {
    if (__hip_atomic_exchange(&sy->counted_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq)
        __hip_atomic_fetch_add(&sy->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void adam_fused_kernel(OptPair o, FusedSync *__restrict__ sy, const unsigned spin_limit, const int test_hold, const unsigned seq) {
    const int k = blockIdx.y, tid = threadIdx.x;
    const OptNet &N = o.net[k];
    const unsigned nblocks = gridDim.x * gridDim.y;
    const int64_t stride = (int64_t)gridDim.x * 256, i0 = (int64_t)blockIdx.x * 256 + tid;
    // ---- phase 1: this workgroup's share of ||g||^2, in double (as sqnorm2_kernel); the first FUSED_EPT elements stay in registers
    float g[FUSED_EPT], m[FUSED_EPT], v[FUSED_EPT], p[FUSED_EPT];
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < FUSED_EPT; ++e) {
        const int64_t i = i0 + e * stride;
        g[e] = i < N.n ? N.g[i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < FUSED_EPT; ++e) {  // requested now, needed after the barrier
        const int64_t i = i0 + e * stride;
        m[e] = i < N.n ? N.m[i] : 0.f;
        v[e] = i < N.n ? N.v[i] : 0.f;
        p[e] = i < N.n ? N.p[i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < FUSED_EPT; ++e) acc += (double)g[e] * (double)g[e];
    for (int64_t i = i0 + FUSED_EPT * stride; i < N.n; i += stride) {
        const double x = (double)N.g[i];
        acc += x * x;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    __shared__ double red[4];
    __shared__ float s_total;
    __shared__ int s_last, s_ok;
    __shared__ unsigned s_g0;
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        int ok = 1, last = 0;
        const unsigned g0 = __hip_atomic_load(&sy->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // read BEFORE arriving
        s_g0 = g0;
        if (g0 == FUSED_DEAD) {  // an earlier launch on this block gave up and its owner has not re-zeroed it: nothing to wait for
            ok = 0;
            if (blockIdx.x == 0 && k == 0) count_skipped_once(sy, seq);
        } else if (test_hold && blockIdx.x == 0 && k == 0) {
            // test hook: this workgroup behaves like one that was never scheduled while the others wait -- it does not arrive; it
            // only watches the generation word until the waiters have given up, so that the launch ends
            while (__hip_atomic_load(&sy->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) __builtin_amdgcn_s_sleep(8);
            ok = 0;
        } else {
            __hip_atomic_store(&sy->slot[k][blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // arrival: the release half orders our slot before the count, the acquire half lets the last arriver see everybody's
            const unsigned old = __hip_atomic_fetch_add(&sy->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = old == nblocks - 1;
            if (!last) {
                unsigned spins = 0, cur;
                while ((cur = __hip_atomic_load(&sy->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == g0) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > spin_limit) {
                        // give up -- unless the barrier opens in this very moment: ONE compare-and-swap decides for the whole launch
                        unsigned expect = g0;
                        if (__hip_atomic_compare_exchange_strong(&sy->gen, &expect, FUSED_DEAD, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT)) {
                            count_skipped_once(sy, seq);
                            cur = FUSED_DEAD;
                        } else {
                            cur = expect;  // somebody else decided first: opened (g0 + 1) or dead
                        }
                        break;
                    }
                }
                ok = cur != FUSED_DEAD;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        s_last = last;
        s_ok = ok;
    }
    __syncthreads();
    if (s_last) {  // the whole workgroup adds the slots: thread t takes slots t, t + 256 of each network; fixed tree afterwards
        __shared__ double part[2][4];
        for (int j = 0; j < 2; ++j) {
            double a = 0.0;
            for (int b = tid; b < (int)gridDim.x; b += 256)
                a += __hip_atomic_load(&sy->slot[j][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) a += __shfl_xor(a, s);
            if ((tid & 63) == 0) part[j][tid >> 6] = a;
        }
        __syncthreads();
        if (tid == 0) {
            for (int j = 0; j < 2; ++j)
                __hip_atomic_store(o.net[j].gnorm2, (part[j][0] + part[j][1]) + (part[j][2] + part[j][3]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // opens the barrier -- unless a waiter gave up first (then nobody updates, this workgroup included)
            unsigned expect = s_g0;
            const unsigned next = s_g0 + 1 == FUSED_DEAD ? 0u : s_g0 + 1;
            s_ok = __hip_atomic_compare_exchange_strong(&sy->gen, &expect, next, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        const double t = __hip_atomic_load(N.gnorm2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_total = (float)sqrt(t);
    }
    __syncthreads();
    if (!s_ok) return;  // given up: parameters, moments and gradients stay as they were (the timeout word says so)
    // ---- phase 2: adam_pack2_kernel, element for element
    const float total = s_total;
    float coef = N.max_norm / (total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
#pragma unroll
    for (int e = 0; e < FUSED_EPT; ++e) {
        const int64_t i = i0 + e * stride;
        if (i < N.n) adam_one(N, i, g[e], m[e], v[e], p[e], coef);
    }
    for (int64_t i = i0 + FUSED_EPT * stride; i < N.n; i += stride) adam_one(N, i, N.g[i], N.m[i], N.v[i], N.p[i], coef);
}

static void fill_jobs(const NetLayout &net, PackJobs *jobs) {
    jobs->n = net.n_layers;
    int64_t tot = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerLayout &L = net.L[l];
        jobs->j[l] = PackJob{L.in, L.out, L.pin, L.pout, L.off_w, L.off_wt, L.off_b, L.off_flat_w, L.off_flat_b, tot};
        tot += (int64_t)L.pin * L.pout;
    }
    jobs->total = tot;
}

int launch_clip_adam_pack2(hipStream_t st, const NetLayout *nets, float *const *p, float *const *g, float *const *m, float *const *v,
                           float *const *packed, double *const *gnorm2, const int64_t *n, const float *max_norm,
                           const float *step_size, const float *bc2_sqrt, const float *omb1, const float *beta2, const float *omb2,
                           const float *eps, void *sync_ws) {
    OptPair o;
    int64_t nmax = 0;
    for (int k = 0; k < 2; ++k) {
        OptNet &N = o.net[k];
        N.p = p[k]; N.g = g[k]; N.m = m[k]; N.v = v[k]; N.packed = packed[k]; N.gnorm2 = gnorm2[k]; N.n = n[k];
        N.max_norm = max_norm[k]; N.step_size = step_size[k]; N.bc2_sqrt = bc2_sqrt[k]; N.omb1 = omb1[k]; N.beta2 = beta2[k];
        N.omb2 = omb2[k]; N.eps = eps[k];
        fill_jobs(nets[k], &N.jobs);
        nmax = n[k] > nmax ? n[k] : nmax;
    }
    if (sync_ws) {  // one launch (the caller owns a zero-initialised sync block)
        RLPPO_CHECK_ARG(((uintptr_t)sync_ws & 15) == 0, "clip_adam_pack2: the sync block must be 16-byte aligned");
        // The grid barrier needs every workgroup of the launch resident at once: per device (queried once for each device id),
        // workgroups per network <= (occupancy x CUs - one workgroup per CU of margin for kernels of other streams) / 2.  On an
        // MI355X that is (8 x 256 - 256) / 2 = 896, far above FUSED_MAX_BLOCKS; a small or partitioned device gets a smaller
        // grid, and one that cannot hold a single workgroup per network the three-operation form below.
        static std::atomic<int> cap_by_dev[64];
        int dev = 0;
        RLPPO_HIP(hipGetDevice(&dev));
        RLPPO_CHECK_ARG(dev >= 0 && dev < 64, "clip_adam_pack2: device id %d", dev);
        int cap = cap_by_dev[dev].load(std::memory_order_acquire);
        if (cap == 0) {
            int per_cu = 0;
            hipDeviceProp_t prop;
            RLPPO_HIP(hipGetDeviceProperties(&prop, dev));
            RLPPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, adam_fused_kernel, 256, 0));
            per_cu = per_cu > 7 ? 7 : per_cu;  // (the occupancy API over-reports by one in the 8-per-CU regime: MI355X_MICROARCH.md)
            cap = (per_cu * prop.multiProcessorCount - prop.multiProcessorCount) / 2;
            cap = cap < 1 ? -1 : cap;
            cap_by_dev[dev].store(cap, std::memory_order_release);
        }
        if (cap >= 1) {
            int64_t blocks = cdiv(nmax, 256 * FUSED_EPT);
            blocks = blocks < 1 ? 1 : (blocks > FUSED_MAX_BLOCKS ? FUSED_MAX_BLOCKS : blocks);
            blocks = blocks > cap ? cap : blocks;
            static std::atomic<unsigned> launch_seq{0};  // per-launch sequence number, never 0 (a zeroed block's counted_seq)
            unsigned seq = launch_seq.fetch_add(1, std::memory_order_relaxed) + 1;
            if (seq == 0) seq = launch_seq.fetch_add(1, std::memory_order_relaxed) + 1;
            hipLaunchKernelGGL(adam_fused_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, st, o, reinterpret_cast<FusedSync *>(sync_ws),
                               g_fused_spin_limit, g_fused_test_hold, seq);
            RLPPO_LAUNCH_CHECK();
            return 0;
        }
    }
    if (gnorm2[1] == gnorm2[0] + 1 || gnorm2[0] == gnorm2[1] + 1) {  // adjacent accumulators (PPOLearner's): one fill
        RLPPO_HIP(hipMemsetAsync(gnorm2[0] < gnorm2[1] ? gnorm2[0] : gnorm2[1], 0, 2 * sizeof(double), st));
    } else {
        for (int k = 0; k < 2; ++k) RLPPO_HIP(hipMemsetAsync(gnorm2[k], 0, sizeof(double), st));
    }
    const int blocks = (int)(cdiv(nmax, 256) < 1024 ? cdiv(nmax, 256) : 1024);
    hipLaunchKernelGGL(sqnorm2_kernel, dim3(blocks, 2), dim3(256), 0, st, o);
    RLPPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_pack2_kernel, dim3(blocks, 2), dim3(256), 0, st, o);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ the report of a learn() [r6]
// ppo_learner.py:213-234 in one launch: REPORT_BLOCKS workgroups sum (before - now)^2 over both flat arenas (float32 differences,
// double squares), each parks its two partial sums in its slot of `ws` and takes a ticket; the last arriver adds the slots in slot
// order (bit-reproducible), writes the statistics, the two norms and the give-up words into the caller's pinned `out`, zeroes the
// device accumulators for the next learn(), re-arms the ticket counter and releases the completion word at system scope.
namespace {
constexpr int REPORT_BLOCKS = 64;
struct ReportWs {
    unsigned tickets;
    unsigned pad[3];
    double partial[REPORT_BLOCKS][2];
};
static_assert(sizeof(ReportWs) <= RLPPO_REPORT_WS_BYTES, "report workspace");
}  // namespace

__global__ __launch_bounds__(256) void learn_report_kernel(rlppo_report_args r) {
    ReportWs *const ws = reinterpret_cast<ReportWs *>(r.ws);
    const int64_t stride = (int64_t)gridDim.x * 256;
    double s[2] = {0.0, 0.0};
    for (int k = 0; k < 2; ++k) {
        const float *b = k ? r.val_before : r.pol_before, *a = k ? r.val_now : r.pol_now;
        const int64_t n = k ? r.n_val : r.n_pol;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float d = b[i] - a[i];
            s[k] += (double)d * (double)d;
        }
    }
    __shared__ double red[4][2];
    __shared__ int last;
    for (int k = 0; k < 2; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws->partial[blockIdx.x][0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        ws->partial[blockIdx.x][1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        __threadfence();
        last = atomicAdd(&ws->tickets, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // the last arriver: its threads fetch the slots side by side (one thread walking 128 dependent loads took 19 of the kernel's 24 us),
    // thread 0 adds them in slot order
    __shared__ double slots[REPORT_BLOCKS][2];
    __threadfence();
    if (threadIdx.x < 2 * gridDim.x)
        slots[threadIdx.x >> 1][threadIdx.x & 1] = __hip_atomic_load(&ws->partial[threadIdx.x >> 1][threadIdx.x & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t0 = 0.0, t1 = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) {
        t0 += slots[b][0];
        t1 += slots[b][1];
    }
    r.stats[RLPPO_STAT_PASSES] += r.add_passes;
    for (int k = 0; k < RLPPO_N_STATS; ++k) {
        r.out[k] = r.stats[k];
        r.stats[k] = 0.0;
    }
    r.out[RLPPO_N_STATS] = sqrt(t0);
    r.out[RLPPO_N_STATS + 1] = sqrt(t1);
    const double own = r.timeout_word ? (double)__hip_atomic_load(r.timeout_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    r.out[RLPPO_N_STATS + 2] = own;
    r.out[RLPPO_N_STATS + 3] = r.extra ? *r.extra : own;
    __hip_atomic_store(&ws->tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(r.done_word, r.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int launch_learn_report(hipStream_t st, const rlppo_report_args &r) {
    hipLaunchKernelGGL(learn_report_kernel, dim3(REPORT_BLOCKS), dim3(256), 0, st, r);
    RLPPO_LAUNCH_CHECK();
    return 0;
}


}  // namespace rlppo
