// gae.hip -- GAE as two segmented reverse affine scans (replaces the Python loop of
// rlgym_ppo/util/torch_functions.py:58-73).
//
// Both recurrences have the form x_t = b_t + a_t * x_{t+1}:
//     advantage:  b_t = clip(r_t/sigma) + gamma*V_{t+1}*nd_t - V_t ,  a_t = gamma*lambda*nd_t*nt_t
//     return:     b_t = r_t ,                                         a_t = gamma*nd_t*nt_t
// with nd = 1 - done, nt = 1 - truncated.  Affine maps compose associatively,
//     (A1,B1) o (A2,B2) = (A1*A2, B1 + A1*B2)      (left segment applied after the right one),
// and a_t = 0 at every trajectory end, so the scan is segmented for free.
//
// HBM-bound (28 algorithmic bytes per step: 16 read, 12 written; ~30 flop).  Two launches:
//   1. gae_summary: every 2048-step block reduces its steps to one composite per recurrence (4 doubles).
//   2. gae_apply:   every block folds the composites of the blocks to its right into its carry-in (stops as soon
//                   as A == 0, i.e. at the first trajectory end -- normally the very next block), re-reads its
//                   steps (served from L2 / Infinity Cache: the 33.5 MB input of the 8192x256 case fits), runs the
//                   in-block suffix scan with wave shuffles and writes the three outputs.
// Arithmetic: float32 reward scaling, float64 recurrences, one rounding to float32 at the store -- the reference's
// behaviour under its pinned NumPy < 2 (oracle/gae_oracle.c mode 0).
#include "common.hpp"

namespace rlppo {

constexpr int GAE_THREADS = 256;
constexpr int GAE_EPT = 8;                           // steps per thread (two float4 per input array)
constexpr int GAE_BLOCK = GAE_THREADS * GAE_EPT;     // 2048 steps per workgroup

struct Aff2 {  // the two composites carried together: advantage (a,b) and return (c,d)
    double a, b, c, d;
};
__device__ __forceinline__ Aff2 aff_identity() { return Aff2{1.0, 0.0, 1.0, 0.0}; }
// left o right
__device__ __forceinline__ Aff2 compose(const Aff2 &l, const Aff2 &r) {
    return Aff2{l.a * r.a, l.b + l.a * r.b, l.c * r.c, l.d + l.c * r.d};
}
__device__ __forceinline__ Aff2 shfl_down_aff(const Aff2 &v, int off) {
    return Aff2{__shfl_down(v.a, off), __shfl_down(v.b, off), __shfl_down(v.c, off), __shfl_down(v.d, off)};
}

struct GaeParams {
    double gamma, gl;  // gamma, gamma*lambda
    float ret_std;
    int use_std;
};

// Loads the GAE_EPT steps of this thread and turns them into per-step coefficients.
struct Steps {
    double a_adv[GAE_EPT], b_adv[GAE_EPT], a_ret[GAE_EPT];
    float r[GAE_EPT], v[GAE_EPT];
};

__device__ __forceinline__ void load_steps(const float *__restrict__ rews, const float *__restrict__ dones,
                                           const float *__restrict__ trunc, const float *__restrict__ values,
                                           int64_t t0, int64_t n, const GaeParams &p, Steps &s) {
    float r[GAE_EPT], d[GAE_EPT], tr[GAE_EPT], v[GAE_EPT + 1];
    if (t0 + GAE_EPT <= n) {
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int h = 0; h < GAE_EPT / 4; ++h) {
            const f4 r4 = *reinterpret_cast<const f4 *>(rews + t0 + 4 * h);
            const f4 d4 = *reinterpret_cast<const f4 *>(dones + t0 + 4 * h);
            const f4 t4 = *reinterpret_cast<const f4 *>(trunc + t0 + 4 * h);
            const f4 v4 = *reinterpret_cast<const f4 *>(values + t0 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[4 * h + e] = r4[e];
                d[4 * h + e] = d4[e];
                tr[4 * h + e] = t4[e];
                v[4 * h + e] = v4[e];
            }
        }
        v[GAE_EPT] = values[t0 + GAE_EPT];
    } else {
#pragma unroll
        for (int e = 0; e < GAE_EPT; ++e) {
            const bool ok = t0 + e < n;
            r[e] = ok ? rews[t0 + e] : 0.f;
            d[e] = ok ? dones[t0 + e] : 0.f;
            tr[e] = ok ? trunc[t0 + e] : 0.f;
            v[e] = ok ? values[t0 + e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < GAE_EPT; ++e)
            if (t0 + e + 1 <= n) v[e + 1] = values[t0 + e + 1];
    }
#pragma unroll
    for (int e = 0; e < GAE_EPT; ++e) {
        if (t0 + e < n) {
            const double nd = (double)(1.0f - d[e]);
            const double nt = (double)(1.0f - tr[e]);
            float rn = r[e];
            if (p.use_std) rn = fminf(fmaxf(r[e] / p.ret_std, -10.f), 10.f);
            s.b_adv[e] = ((double)rn + p.gamma * (double)v[e + 1] * nd) - (double)v[e];
            s.a_adv[e] = p.gl * nd * nt;
            s.a_ret[e] = p.gamma * nd * nt;
        } else {  // past the end: identity, so partial blocks need no special casing downstream
            s.b_adv[e] = 0.0;
            s.a_adv[e] = 1.0;
            s.a_ret[e] = 1.0;
            r[e] = 0.f;
        }
        s.r[e] = r[e];
        s.v[e] = v[e];
    }
}

__device__ __forceinline__ Aff2 thread_composite(const Steps &s) {
    Aff2 c = aff_identity();
#pragma unroll
    for (int e = GAE_EPT - 1; e >= 0; --e) {  // right to left: c <- step_e o c
        c.b = s.b_adv[e] + s.a_adv[e] * c.b;
        c.a = s.a_adv[e] * c.a;
        c.d = (double)s.r[e] + s.a_ret[e] * c.d;
        c.c = s.a_ret[e] * c.c;
    }
    return c;
}

// Ordered reduction of one composite per thread over the workgroup (thread 0 is the leftmost segment).
// Returns the workgroup composite in every thread.
__device__ __forceinline__ Aff2 block_reduce(Aff2 v, Aff2 *lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Aff2 o = shfl_down_aff(v, off);
        if (lane + off < 64) v = compose(v, o);  // lane 0 ends with lanes 0..63 in order
    }
    __syncthreads();
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    return compose(compose(lds4[0], lds4[1]), compose(lds4[2], lds4[3]));
}

__global__ __launch_bounds__(GAE_THREADS) void gae_summary_kernel(const float *__restrict__ rews,
                                                                   const float *__restrict__ dones,
                                                                   const float *__restrict__ trunc,
                                                                   const float *__restrict__ values, int64_t n,
                                                                   GaeParams p, Aff2 *__restrict__ summaries) {
    __shared__ Aff2 lds4[4];
    const int64_t t0 = ((int64_t)blockIdx.x * GAE_THREADS + threadIdx.x) * GAE_EPT;
    Steps s;
    load_steps(rews, dones, trunc, values, t0, n, p, s);
    const Aff2 c = block_reduce(thread_composite(s), lds4);
    if (threadIdx.x == 0) summaries[blockIdx.x] = c;
}

__global__ __launch_bounds__(GAE_THREADS) void gae_apply_kernel(const float *__restrict__ rews,
                                                                 const float *__restrict__ dones,
                                                                 const float *__restrict__ trunc,
                                                                 const float *__restrict__ values, int64_t n,
                                                                 GaeParams p, const Aff2 *__restrict__ summaries,
                                                                 int n_blocks, float *__restrict__ vt_out,
                                                                 float *__restrict__ adv_out,
                                                                 float *__restrict__ ret_out) {
    __shared__ Aff2 lds4[4];
    __shared__ Aff2 wave_tot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t0 = ((int64_t)blockIdx.x * GAE_THREADS + threadIdx.x) * GAE_EPT;

    // issue this block's loads first; the carry look-up below overlaps with them
    Steps s;
    load_steps(rews, dones, trunc, values, t0, n, p, s);

    // ---- carry-in: x at the first step of block+1 = (composite of every block to the right)(0)
    Aff2 right = aff_identity();
    for (int base = blockIdx.x + 1; base < n_blocks; base += GAE_THREADS) {
        const int j = base + threadIdx.x;
        const Aff2 mine = j < n_blocks ? summaries[j] : aff_identity();
        right = compose(right, block_reduce(mine, lds4));
        if (right.a == 0.0 && right.c == 0.0) break;  // uniform: a trajectory end cuts both recurrences
    }
    const double carry_adv = right.b, carry_ret = right.d;

    // ---- in-block exclusive suffix scan of the thread composites
    const Aff2 mine = thread_composite(s);
    Aff2 inc = mine;  // inclusive suffix over lanes lane..63
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Aff2 o = shfl_down_aff(inc, off);
        if (lane + off < 64) inc = compose(inc, o);
    }
    __syncthreads();
    if (lane == 0) wave_tot[wave] = inc;
    __syncthreads();
    Aff2 after = shfl_down_aff(inc, 1);  // lanes lane+1..63
    if (lane == 63) after = aff_identity();
    for (int w = wave + 1; w < 4; ++w) after = compose(after, wave_tot[w]);
    double x_adv = after.b + after.a * carry_adv;  // x_{t+1} entering this thread's rightmost step
    double x_ret = after.d + after.c * carry_ret;

    float o_adv[GAE_EPT], o_vt[GAE_EPT], o_ret[GAE_EPT];
#pragma unroll
    for (int e = GAE_EPT - 1; e >= 0; --e) {
        x_adv = s.b_adv[e] + s.a_adv[e] * x_adv;
        x_ret = (double)s.r[e] + s.a_ret[e] * x_ret;
        o_adv[e] = (float)x_adv;
        o_vt[e] = (float)((double)s.v[e] + x_adv);
        o_ret[e] = (float)x_ret;
    }
    if (t0 + GAE_EPT <= n) {
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int h = 0; h < GAE_EPT / 4; ++h) {
            *reinterpret_cast<f4 *>(adv_out + t0 + 4 * h) = f4{o_adv[4 * h], o_adv[4 * h + 1], o_adv[4 * h + 2], o_adv[4 * h + 3]};
            *reinterpret_cast<f4 *>(vt_out + t0 + 4 * h) = f4{o_vt[4 * h], o_vt[4 * h + 1], o_vt[4 * h + 2], o_vt[4 * h + 3]};
            *reinterpret_cast<f4 *>(ret_out + t0 + 4 * h) = f4{o_ret[4 * h], o_ret[4 * h + 1], o_ret[4 * h + 2], o_ret[4 * h + 3]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < GAE_EPT; ++e)
            if (t0 + e < n) {
                adv_out[t0 + e] = o_adv[e];
                vt_out[t0 + e] = o_vt[e];
                ret_out[t0 + e] = o_ret[e];
            }
    }
}

size_t gae_workspace_bytes(int64_t n) { return (size_t)(cdiv(n > 0 ? n : 1, GAE_BLOCK)) * sizeof(Aff2); }

int launch_gae(hipStream_t st, const float *rews, const float *dones, const float *trunc, const float *values, int64_t n,
               double gamma, double lmbda, float ret_std, float *vt, float *adv, float *ret, void *ws, size_t ws_bytes) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0, "gae: n=%ld", (long)n);
    RLPPO_CHECK_ARG(((uintptr_t)rews | (uintptr_t)dones | (uintptr_t)trunc | (uintptr_t)values | (uintptr_t)vt |
                     (uintptr_t)adv | (uintptr_t)ret) % 16 == 0,
                    "gae: arrays must be 16-byte aligned");
    if (ws_bytes < gae_workspace_bytes(n)) {
        set_error("gae: workspace %zu < %zu bytes", ws_bytes, gae_workspace_bytes(n));
        return RLPPO_ERR_WORKSPACE;
    }
    GaeParams p;
    p.gamma = gamma;
    p.gl = gamma * lmbda;
    p.use_std = !(ret_std != ret_std);  // NaN means "no scaling" (return_std=None)
    p.ret_std = ret_std;
    const int nb = (int)cdiv(n, GAE_BLOCK);
    Aff2 *summ = reinterpret_cast<Aff2 *>(ws);
    hipLaunchKernelGGL(gae_summary_kernel, dim3(nb), dim3(GAE_THREADS), 0, st, rews, dones, trunc, values, n, p, summ);
    RLPPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(gae_apply_kernel, dim3(nb), dim3(GAE_THREADS), 0, st, rews, dones, trunc, values, n, p, summ, nb,
                       vt, adv, ret);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo
