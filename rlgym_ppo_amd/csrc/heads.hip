// heads.hip -- action-head kernels: sampling for rollout inference and the fused loss epilogue of the PPO update.
//
// Discrete head: one wave (64 lanes) per row, EPL logits per lane held in registers, wave-shuffle reductions
// (max, sum-exp, entropy, the softmax-Jacobian dot product, arg-max of p/q).  The chain is the reference's
// literal softmax -> clamp(1e-11, 1) -> log (discrete_policy.py:52-54,70-78), NOT log_softmax, including the
// clamp's zero-gradient region and torch.min's tie rule.
// Gaussian / multi-discrete heads have 16 / 21 outputs per row: one thread per row, everything in registers.
#include "common.hpp"

namespace rlppo {

constexpr float PROB_MIN = 1e-11f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// softmax + clamp of one row spread over a wave; element c = lane + 64 e.
template <int EPL>
__device__ __forceinline__ void row_softmax(const float *__restrict__ z, int A, int lane, float (&p)[EPL],
                                            float (&pc)[EPL]) {
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        p[e] = c < A ? z[c] : -INFINITY;
        mx = fmaxf(mx, p[e]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        p[e] = c < A ? expf(p[e] - mx) : 0.f;
        s += p[e];
    }
    s = wave_sum(s);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        p[e] = p[e] / s;
        pc[e] = fminf(fmaxf(p[e], PROB_MIN), 1.0f);
    }
}

// ----------------------------------------------------------------------------------- discrete: sampling
// action = argmax_c pc[c] / q[c] (first index wins ties), logp = log(pc[action]).  `from_probs`: the row already
// holds clamped probabilities (rlppo_categorical_select).
template <int EPL, bool FROM_PROBS>
__global__ __launch_bounds__(256) void discrete_sample_kernel(const float *__restrict__ src, int64_t ld, int64_t n,
                                                               int A, const float *__restrict__ noise,
                                                               int64_t *__restrict__ actions, float *__restrict__ logp,
                                                               float *__restrict__ probs_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float p[EPL], pc[EPL];
    const float *z = src + row * ld;
    if (FROM_PROBS) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            pc[e] = c < A ? z[c] : 0.f;
        }
    } else {
        row_softmax<EPL>(z, A, lane, p, pc);
    }
    float best = -INFINITY;
    int besti = 0x7fffffff;
    float bestp = 1.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < A) {
            const float v = pc[e] / noise[row * A + c];  // IEEE fp32 division, as at::div
            if (v > best) {
                best = v;
                besti = c;
                bestp = pc[e];
            }
            if (probs_out) probs_out[row * A + c] = pc[e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        const float op = __shfl_xor(bestp, o);
        if (ov > best || (ov == best && oi < besti)) {
            best = ov;
            besti = oi;
            bestp = op;
        }
    }
    if (lane == 0) {
        actions[row] = besti;
        logp[row] = logf(bestp);
    }
}

template <bool FROM_PROBS>
static int launch_discrete_sample(hipStream_t st, const float *src, int64_t ld, int64_t n, int A, const float *noise,
                                  int64_t *actions, float *logp, float *probs_out) {
    if (n <= 0) return 0;
    dim3 grid((unsigned)cdiv(n, 4)), block(256);
    if (A <= 128)
        hipLaunchKernelGGL((discrete_sample_kernel<2, FROM_PROBS>), grid, block, 0, st, src, ld, n, A, noise, actions, logp, probs_out);
    else if (A <= 512)
        hipLaunchKernelGGL((discrete_sample_kernel<8, FROM_PROBS>), grid, block, 0, st, src, ld, n, A, noise, actions, logp, probs_out);
    else if (A <= 2048)
        hipLaunchKernelGGL((discrete_sample_kernel<32, FROM_PROBS>), grid, block, 0, st, src, ld, n, A, noise, actions, logp, probs_out);
    else {
        set_error("discrete head: n_actions=%d > 2048 unsupported", A);
        return RLPPO_ERR_ARG;
    }
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_discrete_sample_logits(hipStream_t st, const float *logits, int64_t ld, int64_t n, int A, const float *noise,
                                  int64_t *actions, float *logp, float *probs_out) {
    return launch_discrete_sample<false>(st, logits, ld, n, A, noise, actions, logp, probs_out);
}
int launch_categorical_select(hipStream_t st, const float *probs, int64_t ld, int64_t n, int A, const float *noise,
                              int64_t *actions, float *logp) {
    return launch_discrete_sample<true>(st, probs, ld, n, A, noise, actions, logp, nullptr);
}

// ---------------------------------------------------------------- discrete: probabilities / deterministic choice
// DiscreteFF.get_output (discrete_policy.py:34-42: softmax) and the deterministic branch of get_action (:52-57: clamp, then
// numpy's argmax over the FLATTENED [n, A] array -- quirk Q11: one index for the whole batch, first occurrence of the maximum).
// The flat arg-max is one 64-bit atomic max per row on key = (float bits of the row maximum << 32) | ~flat index: clamped
// probabilities are positive, so their bit patterns order like the values, and of equal values the smaller flat index wins.
template <int EPL>
__global__ __launch_bounds__(256) void discrete_probs_kernel(const float *__restrict__ logits, int64_t ld, int64_t n, int A,
                                                              int clamp, float *__restrict__ probs, int64_t ld_p,
                                                              unsigned long long *__restrict__ key) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float p[EPL], pc[EPL];
    row_softmax<EPL>(logits + row * ld, A, lane, p, pc);
    float best = -1.f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int c = lane + 64 * e;
        if (c < A) {
            if (probs) probs[row * ld_p + c] = clamp ? pc[e] : p[e];
            if (pc[e] > best) {
                best = pc[e];
                besti = c;
            }
        }
    }
    if (!key) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        if (ov > best || (ov == best && oi < besti)) {
            best = ov;
            besti = oi;
        }
    }
    if (lane == 0) {
        const unsigned long long flat = (unsigned long long)(row * A + besti);
        atomicMax(key, ((unsigned long long)__float_as_uint(best) << 32) | (0xffffffffull - flat));
    }
}
__global__ void decode_flat_argmax_kernel(unsigned long long *key) {
    *(long long *)key = (long long)(0xffffffffull - (*key & 0xffffffffull));
}

int launch_discrete_probs(hipStream_t st, const float *logits, int64_t ld, int64_t n, int A, int clamp, float *probs,
                          int64_t ld_p, int64_t *flat_argmax) {
    if (n <= 0) return 0;
    if (flat_argmax && (unsigned long long)n * (unsigned long long)A > 0xffffffffull) {
        set_error("discrete_probs: n * n_actions exceeds the 32-bit flat index of the arg-max key");
        return RLPPO_ERR_ARG;
    }
    unsigned long long *key = (unsigned long long *)flat_argmax;
    if (key) RLPPO_HIP(hipMemsetAsync(key, 0, sizeof(*key), st));
    dim3 grid((unsigned)cdiv(n, 4)), block(256);
    if (A <= 128)
        hipLaunchKernelGGL((discrete_probs_kernel<2>), grid, block, 0, st, logits, ld, n, A, clamp, probs, ld_p, key);
    else if (A <= 512)
        hipLaunchKernelGGL((discrete_probs_kernel<8>), grid, block, 0, st, logits, ld, n, A, clamp, probs, ld_p, key);
    else if (A <= 2048)
        hipLaunchKernelGGL((discrete_probs_kernel<32>), grid, block, 0, st, logits, ld, n, A, clamp, probs, ld_p, key);
    else {
        set_error("discrete head: n_actions=%d > 2048 unsupported", A);
        return RLPPO_ERR_ARG;
    }
    RLPPO_LAUNCH_CHECK();
    if (key) {
        hipLaunchKernelGGL(decode_flat_argmax_kernel, dim3(1), dim3(1), 0, st, key);
        RLPPO_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------ shared loss pieces
// d/d(ratio) of min(ratio*A, clamp(ratio)*A), divided by A
__device__ __forceinline__ float surrogate_weight(float ratio, float adv, const LossCfg &c, float &s_min) {
    const float s1 = ratio * adv;
    const float cl = fminf(fmaxf(ratio, c.clip_lo), c.clip_hi);
    const float s2 = cl * adv;
    const float inr = (ratio >= c.clip_lo && ratio <= c.clip_hi) ? 1.f : 0.f;
    s_min = fminf(s1, s2);
    return s1 < s2 ? 1.f : (s1 > s2 ? inr : 0.5f + 0.5f * inr);
}

// block-level accumulation of per-row statistics into the double accumulators
__device__ __forceinline__ void block_stats_add(double *stats, const float (&v)[5], bool active) {
    __shared__ float red[5][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float x = wave_sum(active ? v[k] : 0.f);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        const double s = (double)red[k][0] + (double)red[k][1] + (double)red[k][2] + (double)red[k][3];
        atomicAdd(stats + k, s);
    }
}

// value loss for one row: writes dL/dv in place, returns (v - t)^2
__device__ __forceinline__ float value_row(float *vout_row, float target, const LossCfg &c) {
    const float v = vout_row[0];
    const float d = v - target;
    vout_row[0] = c.mb_ratio * (2.f * d * c.inv_mb);
    return d * d;
}

// ----------------------------------------------------------------------------------- discrete: loss + grad
// One wave per row.  logits[row][0:A] is overwritten with dL/dlogits, vout[row][0] with dL/dv.
template <int EPL>
__global__ __launch_bounds__(256) void discrete_loss_kernel(float *__restrict__ logits, int64_t ld, int A,
                                                             float *__restrict__ vout, int64_t ldv,
                                                             const int64_t *__restrict__ idx,
                                                             const float *__restrict__ actions,
                                                             const float *__restrict__ old_logp,
                                                             const float *__restrict__ targets,
                                                             const float *__restrict__ advantages, int64_t mb,
                                                             LossCfg cfg, double *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // grid-stride over rows: statistics stay in registers, so a launch issues 5 atomics per BLOCK, not per 4 rows
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < mb; row += (int64_t)gridDim.x * 4) {
        const int64_t src = idx ? ring_row(idx[row], cfg.ring_base, cfg.ring_cap) : row;  // idx == null: per-row data already gathered
        float *z = logits + row * ld;
        float p[EPL], pc[EPL], lp[EPL];
        row_softmax<EPL>(z, A, lane, p, pc);
        const int a = (int)actions[src];  // acts.long() of a float-encoded index (discrete_policy.py:71)
        float ent = 0.f, lpa = 0.f, pca = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            lp[e] = c < A ? logf(pc[e]) : 0.f;
            if (c < A) ent -= lp[e] * pc[e];
            if (c == a) {
                lpa = lp[e];
                pca = pc[e];
            }
        }
        ent = wave_sum(ent);
        lpa = wave_sum(lpa);  // exactly one lane holds a non-zero term
        pca = wave_sum(pca);
        const float old = old_logp[src], adv = advantages[src];
        const float lr = lpa - old;
        const float ratio = expf(lr);
        float smin;
        const float w = surrogate_weight(ratio, adv, cfg, smin);
        const float g_logp = cfg.mb_ratio * (-(adv * w * ratio) * cfg.inv_mb);
        const float g_ent = cfg.mb_ratio * (cfg.ent_coef * cfg.inv_mb);
        float gp[EPL];
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            float g = 0.f;
            if (c < A) {
                g = g_ent * (lp[e] + 1.f);
                if (c == a) g += g_logp / pca;
                if (!(p[e] >= PROB_MIN)) g = 0.f;  // clamp passes gradient on [1e-11, 1] only
            }
            gp[e] = g;
            dot += g * p[e];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int c = lane + 64 * e;
            if (c < ld) z[c] = c < A ? p[e] * (gp[e] - dot) : 0.f;
        }
        if (lane == 0) {
            st[RLPPO_STAT_ENTROPY] += ent * cfg.inv_mb;
            st[RLPPO_STAT_KL] += ((ratio - 1.f) - lr) * cfg.inv_mb;
            st[RLPPO_STAT_CLIPFRAC] += (fabsf(ratio - 1.f) > cfg.clip ? 1.f : 0.f) * cfg.inv_mb;
            st[RLPPO_STAT_PLOSS] += -smin * cfg.inv_mb;
            if (vout) st[RLPPO_STAT_VLOSS] += value_row(vout + row * ldv, targets[src], cfg) * cfg.inv_mb;
        }
    }
    block_stats_add(stats, st, true);
}

// ---- 16 lanes per row (padded width <= 128): a wave works on 4 rows at once, a lane holds 8 CONSECUTIVE logits (two
// 16-byte loads/stores) and the six per-row reductions are 4 DPP steps inside a 16-lane row instead of 6 cross-lane
// shuffles of a whole wave.  Same arithmetic chain as discrete_loss_kernel; only the summation order inside a row differs.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// all-reduce over a 16-lane DPP row: xor 1, xor 2 (quad_perm), then half-row mirror and row mirror (every lane of a quad /
// half row already holds the same partial, so a mirror is as good as an xor)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}

__global__ __launch_bounds__(256) void discrete_loss16_kernel(float *__restrict__ logits, int64_t ld, int A,
                                                               float *__restrict__ vout, int64_t ldv,
                                                               const int64_t *__restrict__ idx,
                                                               const float *__restrict__ actions,
                                                               const float *__restrict__ old_logp,
                                                               const float *__restrict__ targets,
                                                               const float *__restrict__ advantages, int64_t mb,
                                                               LossCfg cfg, double *__restrict__ stats) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c0 = l16 * 8;
    const bool in_row = c0 < ld;  // ld is a multiple of 32, so a lane's 8 columns are all inside or all outside
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t base = (int64_t)blockIdx.x * 16; base < mb; base += (int64_t)gridDim.x * 16) {
        const bool live = base + grp < mb;
        const int64_t row = live ? base + grp : mb - 1;  // idle groups of the last pass shadow the last row (no stores)
        const int64_t src = idx ? ring_row(idx[row], cfg.ring_base, cfg.ring_cap) : row;  // idx == null: per-row data already gathered
        float *z = logits + row * ld + c0;
        float p[8], pc[8], lp[8];
        if (in_row) {
            const f32x4 z0 = *reinterpret_cast<const f32x4 *>(z), z1 = *reinterpret_cast<const f32x4 *>(z + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p[e] = z0[e];
                p[4 + e] = z1[e];
            }
        }
        const float old = old_logp[src], adv = advantages[src];
        const int a = (int)actions[src];  // acts.long() of a float-encoded index (discrete_policy.py:71)
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!(in_row && c0 + e < A)) p[e] = -INFINITY;
            mx = fmaxf(mx, p[e]);
        }
        mx = row16_max(mx);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = (c0 + e < A) ? expf(p[e] - mx) : 0.f;
            s += p[e];
        }
        s = row16_sum(s);
        float ent = 0.f, lpa = 0.f, pca = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            p[e] = p[e] / s;
            pc[e] = fminf(fmaxf(p[e], PROB_MIN), 1.0f);
            lp[e] = c < A ? logf(pc[e]) : 0.f;
            if (c < A) ent -= lp[e] * pc[e];
            if (c == a) {
                lpa = lp[e];
                pca = pc[e];
            }
        }
        ent = row16_sum(ent);
        lpa = row16_sum(lpa);  // exactly one lane holds a non-zero term
        pca = row16_sum(pca);
        const float lr = lpa - old;
        const float ratio = expf(lr);
        float smin;
        const float w = surrogate_weight(ratio, adv, cfg, smin);
        const float g_logp = cfg.mb_ratio * (-(adv * w * ratio) * cfg.inv_mb);
        const float g_ent = cfg.mb_ratio * (cfg.ent_coef * cfg.inv_mb);
        float gp[8];
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float g = 0.f;
            if (c < A) {
                g = g_ent * (lp[e] + 1.f);
                if (c == a) g += g_logp / pca;
                if (!(p[e] >= PROB_MIN)) g = 0.f;  // clamp passes gradient on [1e-11, 1] only
            }
            gp[e] = g;
            dot += g * p[e];
        }
        dot = row16_sum(dot);
        if (live && in_row) {
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = (c0 + e < A) ? p[e] * (gp[e] - dot) : 0.f;
                o1[e] = (c0 + 4 + e < A) ? p[4 + e] * (gp[4 + e] - dot) : 0.f;
            }
            *reinterpret_cast<f32x4 *>(z) = o0;
            *reinterpret_cast<f32x4 *>(z + 4) = o1;
        }
        if (live && l16 == 0) {
            st[RLPPO_STAT_ENTROPY] += ent * cfg.inv_mb;
            st[RLPPO_STAT_KL] += ((ratio - 1.f) - lr) * cfg.inv_mb;
            st[RLPPO_STAT_CLIPFRAC] += (fabsf(ratio - 1.f) > cfg.clip ? 1.f : 0.f) * cfg.inv_mb;
            st[RLPPO_STAT_PLOSS] += -smin * cfg.inv_mb;
            if (vout) st[RLPPO_STAT_VLOSS] += value_row(vout + row * ldv, targets[src], cfg) * cfg.inv_mb;
        }
    }
    block_stats_add(stats, st, true);
}

// Value loss on its own (value_estimator + ppo_learner.py:163-166: MSE(vals, target_values)): v -> d loss / d v in place,
// VLOSS statistic.  A separate launch so that the critic's launch chain never has to meet the policy's between the
// forward and the backward pass: the two chains only join at the end of the minibatch.
__global__ __launch_bounds__(256) void value_loss_kernel(float *__restrict__ vout, int64_t ldv, const int64_t *__restrict__ idx,
                                                         const float *__restrict__ targets, int64_t mb, LossCfg cfg,
                                                         double *__restrict__ stats) {
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < mb; row += (int64_t)gridDim.x * 256)
        st[RLPPO_STAT_VLOSS] += value_row(vout + row * ldv, targets[idx ? ring_row(idx[row], cfg.ring_base, cfg.ring_cap) : row], cfg) * cfg.inv_mb;
    block_stats_add(stats, st, true);
}

int launch_value_loss(hipStream_t st, float *vout, int64_t ldv, const int64_t *idx, const float *targets, int64_t mb,
                      const LossCfg &cfg, double *stats) {
    if (mb <= 0) return 0;
    dim3 grid((unsigned)(cdiv(mb, 256) < 1024 ? cdiv(mb, 256) : 1024));
    hipLaunchKernelGGL(value_loss_kernel, grid, dim3(256), 0, st, vout, ldv, idx, targets, mb, cfg, stats);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_discrete_loss(hipStream_t st, float *logits, int64_t ld, int A, float *vout, int64_t ldv, const int64_t *idx,
                         const float *actions, const float *old_logp, const float *targets, const float *adv, int64_t mb,
                         const LossCfg &cfg, double *stats) {
    if (mb <= 0) return 0;
    dim3 grid((unsigned)(cdiv(mb, 4) < 2048 ? cdiv(mb, 4) : 2048)), block(256);
    RLPPO_CHECK_ARG(ld <= 64 * 32, "discrete head: padded width %ld too large", (long)ld);
    if (ld <= 128) {  // 16 lanes per row, DPP reductions
        dim3 grid16((unsigned)(cdiv(mb, 16) < 2048 ? cdiv(mb, 16) : 2048));
        hipLaunchKernelGGL(discrete_loss16_kernel, grid16, block, 0, st, logits, ld, A, vout, ldv, idx, actions, old_logp, targets, adv, mb, cfg, stats);
    } else if (ld <= 512)  // one wave per row
        hipLaunchKernelGGL((discrete_loss_kernel<8>), grid, block, 0, st, logits, ld, A, vout, ldv, idx, actions, old_logp, targets, adv, mb, cfg, stats);
    else
        hipLaunchKernelGGL((discrete_loss_kernel<32>), grid, block, 0, st, logits, ld, A, vout, ldv, idx, actions, old_logp, targets, adv, mb, cfg, stats);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// --------------------------------------------------------------------------------------------- gaussian
constexpr int MAX_K = 32;  // action dimensions supported by the one-thread-per-row kernels

__device__ __forceinline__ float gauss_logpdf(float x, float mean, float sd) {
    // the reference's four terms, in its order (continuous_policy.py:54-63)
    const float msq = mean * mean, ssq = sd * sd, xsq = x * x;
    const float t1 = -(msq / (2.f * ssq));
    const float t2 = (mean * x) / ssq;
    const float t3 = -(xsq / (2.f * ssq));
    const float t4 = logf(1.f / sqrtf((float)(2.0 * 3.14159265358979323846) * ssq));
    return t1 + t2 + t3 + t4;
}

// [r5] completion words of a sampling kernel that gives a block 256 whole rows = the 16 words 16 b .. 16 b + 15 (rlppo_act_opts.done_words):
// every thread releases its stores at system scope, a barrier, sixteen stores -- the words ride in the call's last kernel instead of
// a launch of their own behind it (one graph node fewer of the small call's eight)
__device__ __forceinline__ void block_done_words(unsigned *done_words, unsigned done_value, int64_t n) {
    if (!done_words) return;  // (uniform)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    const int64_t w = (int64_t)blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x < 16 && w < (n + 15) / 16) __hip_atomic_store(done_words + w, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// y[row][0:2k] holds tanh outputs.  action = clamp(mean + sd*eps, -1, 1); logp = sum logpdf(action)
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float *__restrict__ y, int64_t ld, int64_t n, int k,
                                                               const float *__restrict__ eps, float var_m, float var_b,
                                                               float *__restrict__ actions, float *__restrict__ logp,
                                                               unsigned *done_words, unsigned done_value) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row < n) {
        const float *yr = y + row * ld;
        float lp = 0.f;
        for (int j = 0; j < k; ++j) {
            const float mean = yr[j];
            const float sd = yr[k + j] * var_m + var_b;
            float a = eps[row * k + j] * sd + mean;  // at::normal: output.mul_(std).add_(mean)
            a = fminf(fmaxf(a, -1.f), 1.f);
            actions[row * k + j] = a;
            lp += gauss_logpdf(a, mean, sd);
        }
        logp[row] = lp;
    }
    block_done_words(done_words, done_value, n);
}

int launch_gaussian_sample(hipStream_t st, const float *y, int64_t ld, int64_t n, int k, const float *eps, float var_m,
                           float var_b, float *actions, float *logp, unsigned *done_words, unsigned done_value) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, y, ld, n, k, eps, var_m,
                       var_b, actions, logp, done_words, done_value);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// y[row][0:2k] (tanh outputs) is overwritten with dL/d(pre-tanh) ; entropy = mean over ALL mb*k elements (quirk Q8)
__global__ __launch_bounds__(256) void gaussian_loss_kernel(float *__restrict__ y, int64_t ld, int k,
                                                             float *__restrict__ vout, int64_t ldv,
                                                             const int64_t *__restrict__ idx,
                                                             const float *__restrict__ actions,
                                                             const float *__restrict__ old_logp,
                                                             const float *__restrict__ targets,
                                                             const float *__restrict__ advantages, int64_t mb,
                                                             LossCfg cfg, double *__restrict__ stats) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = row < mb;
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int64_t src = idx ? ring_row(idx[row], cfg.ring_base, cfg.ring_cap) : row;  // idx == null: per-row data already gathered
        float *yr = y + row * ld;
        float mean[MAX_K], sd[MAX_K], x[MAX_K];
        float lp = 0.f, ent = 0.f;
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            mean[j] = yr[j];
            sd[j] = yr[k + j] * cfg.var_m + cfg.var_b;
            x[j] = actions[src * k + j];
            lp += gauss_logpdf(x[j], mean[j], sd[j]);
            ent += 1.4189385332046727f + logf(sd[j]);  // 0.5 + 0.5*log(2*pi) + log(sd): Normal.entropy()
        }
        const float old = old_logp[src], adv = advantages[src];
        const float lr = lp - old;
        const float ratio = expf(lr);
        float smin;
        const float w = surrogate_weight(ratio, adv, cfg, smin);
        const float g_logp = cfg.mb_ratio * (-(adv * w * ratio) * cfg.inv_mb);
        const float g_ent = -cfg.mb_ratio * cfg.ent_coef * cfg.inv_mb / (float)k;  // d(-c_H * H)/d log sd
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            const float d = x[j] - mean[j];
            const float s2 = sd[j] * sd[j];
            const float d_mu = g_logp * d / s2;
            const float d_sd = g_logp * (d * d / (s2 * sd[j]) - 1.f / sd[j]) + g_ent / sd[j];
            const float ym = yr[j], ys = yr[k + j];
            yr[j] = d_mu * (1.f - ym * ym);
            yr[k + j] = d_sd * cfg.var_m * (1.f - ys * ys);
        }
        st[RLPPO_STAT_ENTROPY] = ent * cfg.inv_mb / (float)k;
        st[RLPPO_STAT_KL] = ((ratio - 1.f) - lr) * cfg.inv_mb;
        st[RLPPO_STAT_CLIPFRAC] = (fabsf(ratio - 1.f) > cfg.clip ? 1.f : 0.f) * cfg.inv_mb;
        st[RLPPO_STAT_PLOSS] = -smin * cfg.inv_mb;
        st[RLPPO_STAT_VLOSS] = vout ? value_row(vout + row * ldv, targets[src], cfg) * cfg.inv_mb : 0.f;
    }
    block_stats_add(stats, st, active);
}

int launch_gaussian_loss(hipStream_t st, float *y, int64_t ld, int k, float *vout, int64_t ldv, const int64_t *idx,
                         const float *actions, const float *old_logp, const float *targets, const float *adv, int64_t mb,
                         const LossCfg &cfg, double *stats) {
    if (mb <= 0) return 0;
    RLPPO_CHECK_ARG(k >= 1 && k <= MAX_K, "gaussian head: action dim %d not in [1, %d]", k, MAX_K);
    hipLaunchKernelGGL(gaussian_loss_kernel, dim3((unsigned)cdiv(mb, 256)), dim3(256), 0, st, y, ld, k, vout, ldv, idx,
                       actions, old_logp, targets, adv, mb, cfg, stats);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------- multi-discrete
// 21 logits = 5 heads of 3 + 3 heads of 2 (multi_discrete_policy.py:20, torch_functions.py:101-113).
__device__ __forceinline__ int md_start(int h) { return h < 5 ? 3 * h : 15 + 2 * (h - 5); }
__device__ __forceinline__ int md_bins(int h) { return h < 5 ? 3 : 2; }

// Categorical(logits).sample() on the [n*8, 3] probability matrix == argmax(p / q) with q[n*8][3]; the padded third
// slot of a 2-way head has p = 0 and never wins.  logp = sum_h log_softmax(z_h)[a_h].
__global__ __launch_bounds__(256) void multidiscrete_sample_kernel(const float *__restrict__ logits, int64_t ld,
                                                                    int64_t n, const float *__restrict__ noise,
                                                                    int64_t *__restrict__ actions,
                                                                    float *__restrict__ logp, unsigned *done_words,
                                                                    unsigned done_value) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (one barrier site for the whole block: a wave whose lanes split at `row < n` would otherwise arrive at the barrier of
    // block_done_words twice -- and release it before its live lanes have stored)
    if (row < n) {
    const float *z = logits + row * ld;
    float lp = 0.f;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const int s = md_start(h), b = md_bins(h);
        float mx = z[s];
        for (int c = 1; c < b; ++c) mx = fmaxf(mx, z[s + c]);
        float e[3], sum = 0.f;
        for (int c = 0; c < b; ++c) {
            e[c] = expf(z[s + c] - mx);
            sum += e[c];
        }
        const float lse = mx + logf(sum);
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < b; ++c) {
            const float v = (e[c] / sum) / noise[(row * 8 + h) * 3 + c];
            if (v > best) {
                best = v;
                bi = c;
            }
        }
        actions[row * 8 + h] = bi;
        lp += z[s + bi] - lse;
    }
    logp[row] = lp;
    }
    block_done_words(done_words, done_value, n);
}

int launch_multidiscrete_sample(hipStream_t st, const float *logits, int64_t ld, int64_t n, const float *noise,
                                int64_t *actions, float *logp, unsigned *done_words, unsigned done_value) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(multidiscrete_sample_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, logits, ld, n, noise,
                       actions, logp, done_words, done_value);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void multidiscrete_loss_kernel(float *__restrict__ logits, int64_t ld,
                                                                  float *__restrict__ vout, int64_t ldv,
                                                                  const int64_t *__restrict__ idx,
                                                                  const float *__restrict__ actions,
                                                                  const float *__restrict__ old_logp,
                                                                  const float *__restrict__ targets,
                                                                  const float *__restrict__ advantages, int64_t mb,
                                                                  LossCfg cfg, double *__restrict__ stats) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = row < mb;
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int64_t src = idx ? ring_row(idx[row], cfg.ring_base, cfg.ring_cap) : row;  // idx == null: per-row data already gathered
        float *z = logits + row * ld;
        float ls[21], ph[21], eh[8];
        int act[8];
        float lp = 0.f, ent = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int s = md_start(h), b = md_bins(h);
            float mx = z[s];
            for (int c = 1; c < b; ++c) mx = fmaxf(mx, z[s + c]);
            float sum = 0.f;
            for (int c = 0; c < b; ++c) sum += expf(z[s + c] - mx);
            const float lse = mx + logf(sum);
            float e = 0.f;
            for (int c = 0; c < b; ++c) {
                ls[s + c] = z[s + c] - lse;
                ph[s + c] = expf(ls[s + c]);
                e -= ph[s + c] * ls[s + c];
            }
            eh[h] = e;
            ent += e;
            act[h] = (int)actions[src * 8 + h];
            lp += ls[s + act[h]];
        }
        const float old = old_logp[src], adv = advantages[src];
        const float lr = lp - old;
        const float ratio = expf(lr);
        float smin;
        const float w = surrogate_weight(ratio, adv, cfg, smin);
        const float g_logp = cfg.mb_ratio * (-(adv * w * ratio) * cfg.inv_mb);
        const float g_ent = -cfg.mb_ratio * cfg.ent_coef * cfg.inv_mb;  // coefficient of d(entropy_row)/dz
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int s = md_start(h), b = md_bins(h);
            for (int c = 0; c < b; ++c) {
                const float onehot = (c == act[h]) ? 1.f : 0.f;
                z[s + c] = g_logp * (onehot - ph[s + c]) + g_ent * (-ph[s + c] * (ls[s + c] + eh[h]));
            }
        }
        for (int c = 21; c < ld; ++c) z[c] = 0.f;
        st[RLPPO_STAT_ENTROPY] = ent * cfg.inv_mb;
        st[RLPPO_STAT_KL] = ((ratio - 1.f) - lr) * cfg.inv_mb;
        st[RLPPO_STAT_CLIPFRAC] = (fabsf(ratio - 1.f) > cfg.clip ? 1.f : 0.f) * cfg.inv_mb;
        st[RLPPO_STAT_PLOSS] = -smin * cfg.inv_mb;
        st[RLPPO_STAT_VLOSS] = vout ? value_row(vout + row * ldv, targets[src], cfg) * cfg.inv_mb : 0.f;
    }
    block_stats_add(stats, st, active);
}

int launch_multidiscrete_loss(hipStream_t st, float *logits, int64_t ld, float *vout, int64_t ldv, const int64_t *idx,
                              const float *actions, const float *old_logp, const float *targets, const float *adv,
                              int64_t mb, const LossCfg &cfg, double *stats) {
    if (mb <= 0) return 0;
    hipLaunchKernelGGL(multidiscrete_loss_kernel, dim3((unsigned)cdiv(mb, 256)), dim3(256), 0, st, logits, ld, vout, ldv,
                       idx, actions, old_logp, targets, adv, mb, cfg, stats);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo
