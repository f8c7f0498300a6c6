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
#include <atomic>
#include <chrono>

#include "common.hpp"

namespace rlppo {

constexpr int GAE_THREADS = 256;
#ifndef GAE_EPT_V
#define GAE_EPT_V 8
#endif
constexpr int GAE_EPT = GAE_EPT_V;                   // steps per thread (GAE_EPT / 4 float4 per input array)
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

// ---- wave-wide inclusive SUFFIX scan of composites: lane l <- f_l o f_{l+1} o ... o f_63 (lane 0 = the wave total).
// The shuffle form (6 steps x 8 ds_bpermute per composite) took ~2.5k cycles per scan and there are two scans on every
// workgroup's critical path (in-kernel stamps: DESIGN.md section 5, GAE history).  Here the four steps inside a 16-lane
// DPP row use row_shl moves (lanes without a source take the identity) and the three row totals to the right of a row
// are fetched with v_readlane: ~56 cheap cross-lane moves instead of 48 LDS-crossbar round trips.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v, double ident) {
    const long long b = __double_as_longlong(v), ib = __double_as_longlong(ident);
    const int lo = __builtin_amdgcn_update_dpp((int)ib, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(ib >> 32), (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ Aff2 dpp_aff(const Aff2 &v) {  // the composite CTRL lanes to the right in the row, or identity
    return Aff2{dpp_f64<CTRL>(v.a, 1.0), dpp_f64<CTRL>(v.b, 0.0), dpp_f64<CTRL>(v.c, 1.0), dpp_f64<CTRL>(v.d, 0.0)};
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ Aff2 readlane_aff(const Aff2 &v, int lane) {
    return Aff2{readlane_f64(v.a, lane), readlane_f64(v.b, lane), readlane_f64(v.c, lane), readlane_f64(v.d, lane)};
}
__device__ __forceinline__ Aff2 wave_suffix_scan(Aff2 v, int lane) {
    v = compose(v, dpp_aff<0x101>(v));  // row_shl:1
    v = compose(v, dpp_aff<0x102>(v));  // row_shl:2
    v = compose(v, dpp_aff<0x104>(v));  // row_shl:4
    v = compose(v, dpp_aff<0x108>(v));  // row_shl:8 -> lane 16 r holds the total of row r
    const Aff2 t1 = readlane_aff(v, 16), t2 = readlane_aff(v, 32), t3 = readlane_aff(v, 48);
    const Aff2 t23 = compose(t2, t3), t123 = compose(t1, t23);
    const int row = lane >> 4;
    const Aff2 right = row == 0 ? t123 : (row == 1 ? t23 : (row == 2 ? t3 : aff_identity()));
    return compose(v, right);
}

struct GaeParams {
    double gamma, gl;  // gamma, gamma*lambda
    float ret_std;
    int use_std;
};

// Loads the GAE_EPT steps of this thread and turns them into per-step coefficients.
struct Steps {
    double b_adv[GAE_EPT];
    // nd * nt of every step is 0 or 1 for 0/1 flags: ONE bit each (bit e); a_adv = gamma*lambda*m and a_ret = gamma*m are formed
    // on use with one select.  (Eight floats here were what pushed the single-pass kernel 3 registers over its 128-VGPR budget: 12
    // bytes of scratch per lane = 3.1 MB of extra HBM writes per 8192 x 256 scan.)  The reference's flags are exactly 0.0 / 1.0
    // (batched_agent_manager.py:145, experience_buffer.py:47-48); launch_gae's contract says so.
    // [r4] A step past the end of the data is the ZERO map (a = 0, b = 0, r = 0: bit clear), not the identity: x = 0 beyond the
    // last step either way, and the coefficient selects lose a level (rounds 1-3 kept a second marker bit per step: 12 more
    // vector instructions per step and use on the path of every full chunk).
    unsigned mbits;
    float r[GAE_EPT], v[GAE_EPT];
};

struct RawSteps {
    float r[GAE_EPT], d[GAE_EPT], tr[GAE_EPT], v[GAE_EPT + 1];
};

// the loads only (so that a caller can put independent work between issuing them and consuming them)
__device__ __forceinline__ void load_raw(const float *__restrict__ rews, const float *__restrict__ dones,
                                         const float *__restrict__ trunc, const float *__restrict__ values, int64_t t0,
                                         int64_t n, RawSteps &w) {
    float *r = w.r, *d = w.d, *tr = w.tr, *v = w.v;
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
        v[GAE_EPT] = 0.f;
#pragma unroll
        for (int e = 0; e < GAE_EPT; ++e)
            if (t0 + e + 1 <= n) v[e + 1] = values[t0 + e + 1];
    }
}

__device__ __forceinline__ void make_steps(const RawSteps &w, int64_t t0, int64_t n, const GaeParams &p, Steps &s) {
    const float *r = w.r, *d = w.d, *tr = w.tr, *v = w.v;
    s.mbits = 0;
#pragma unroll
    for (int e = 0; e < GAE_EPT; ++e) {
        if (t0 + e < n) {
            const float ndf = 1.0f - d[e], ntf = 1.0f - tr[e];
            float rn = r[e];
            if (p.use_std) rn = fminf(fmaxf(r[e] / p.ret_std, -10.f), 10.f);
            s.b_adv[e] = ((double)rn + p.gamma * (double)v[e + 1] * (double)ndf) - (double)v[e];
            s.mbits |= (ndf * ntf != 0.f ? 1u : 0u) << e;  // (0/1 flags: the float product is the exact 0/1 the reference's double one is)
            s.r[e] = r[e];
        } else {  // past the end: the zero map
            s.b_adv[e] = 0.0;
            s.r[e] = 0.f;
        }
        s.v[e] = v[e];
    }
}

__device__ __forceinline__ void load_steps(const float *__restrict__ rews, const float *__restrict__ dones,
                                           const float *__restrict__ trunc, const float *__restrict__ values,
                                           int64_t t0, int64_t n, const GaeParams &p, Steps &s) {
    RawSteps w;
    load_raw(rews, dones, trunc, values, t0, n, w);
    make_steps(w, t0, n, p, s);
}

__device__ __forceinline__ double coef_adv(const Steps &s, int e, const GaeParams &p) { return (s.mbits >> e) & 1u ? p.gl : 0.0; }
__device__ __forceinline__ double coef_ret(const Steps &s, int e, const GaeParams &p) { return (s.mbits >> e) & 1u ? p.gamma : 0.0; }

__device__ __forceinline__ Aff2 thread_composite(const Steps &s, const GaeParams &p) {
    Aff2 c = aff_identity();
#pragma unroll
    for (int e = GAE_EPT - 1; e >= 0; --e) {  // right to left: c <- step_e o c
        const double aa = coef_adv(s, e, p), ar = coef_ret(s, e, p);
        c.b = s.b_adv[e] + aa * c.b;
        c.a = aa * c.a;
        c.d = (double)s.r[e] + ar * c.d;
        c.c = ar * c.c;
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
    const Aff2 c = block_reduce(thread_composite(s, p), lds4);
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
    const Aff2 mine = thread_composite(s, p);
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
        x_adv = s.b_adv[e] + coef_adv(s, e, p) * x_adv;
        x_ret = (double)s.r[e] + coef_ret(s, e, p) * x_ret;
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

// one raw step as its pair of affine maps (the look-ahead window and the slow path below work on raw steps)
__device__ __forceinline__ Aff2 step_affine(float r, float d, float t, float v, float v1, const GaeParams &p) {
    const double nd = (double)(1.0f - d), nt = (double)(1.0f - t);
    const float rn = p.use_std ? fminf(fmaxf(r / p.ret_std, -10.f), 10.f) : r;
    const double m = (double)(float)(nd * nt);
    return Aff2{p.gl * m, ((double)rn + p.gamma * (double)v1 * nd) - (double)v, p.gamma * m, (double)r};
}

// ------------------------------------------------------------------------------------------ single pass
// One launch: chained scan with decoupled look-back.  Workgroups take chunks right-to-left through a ticket counter
// (a chunk only ever waits on chunks whose tickets were drawn earlier, so progress never depends on dispatch order or
// residency), publish their composite, look at the chunks to their right until an inclusive value or A == 0 is
// found (normally one hop), publish their own inclusive value and finish the in-block scan.  Every published word
// travels in an 8-byte {tag, value} granule written by ONE agent-scope relaxed atomic store (write-through, sc1) and
// read by agent-scope relaxed atomic loads that bypass L1: the data is its own flag, no fences
// (cdna_hip_programming.md Guideline 16, form R2).  No memset: tags are per-launch epochs (see the kernel).
typedef unsigned long long u64;
constexpr int LB_AGG = 8;                       // granules of the aggregate record: 4 doubles
constexpr int LB_INC = 4;                       // granules of the inclusive record: 2 doubles
constexpr int LB_STRIDE = 16;                   // granules per chunk (128 B: one line per chunk)
constexpr int LOOKAHEAD = 256;                  // raw steps of the next chunk inspected by wave 0 (4 per lane)
constexpr unsigned LB_SPIN_LIMIT = 1u << 20;    // bounded spin (~1 s): never a hang; a timeout poisons the outputs with NaN

__device__ __forceinline__ void put_granule(u64 *p, unsigned tag, unsigned v) {
    __hip_atomic_store(p, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put_double(u64 *p, unsigned tag, double d) {
    const u64 bits = (u64)__double_as_longlong(d);
    put_granule(p, tag, (unsigned)bits);
    put_granule(p + 1, tag, (unsigned)(bits >> 32));
}

// Wave 0 of a workgroup resolves the chunk's carry-in (x at the first step of chunk + 1) and leaves it in s_carry: the look-ahead
// fast path over the window composites in wave_la, else the decoupled look-back over published records.  Shared by both kernel forms.
__device__ __forceinline__ void resolve_carry(const int chunk, const int n_blocks, const int64_t n, const int lane, const int wave,
                                              const Aff2 *wave_la, const Aff2 &agg, const bool prefix_open, u64 *rec,
                                              u64 *__restrict__ state, const unsigned TAG_AGG, const unsigned TAG_INC,
                                              const unsigned spin_limit, unsigned *__restrict__ slow_word, double *s_carry) {
    if (wave == 0) {
        double carry_adv = 0.0, carry_ret = 0.0;
        if (chunk + 1 < n_blocks) {
            // Fast path: compose the first LOOKAHEAD raw steps of the next chunk.  A trajectory end inside them
            // (a == 0) fixes the carry with no dependence on any other workgroup -- the normal case for rollout data.
            Aff2 acc = aff_identity();
            bool done = false;
            {
                const Aff2 la = compose(compose(wave_la[0], wave_la[1]), compose(wave_la[2], wave_la[3]));
                const bool covers_all = (int64_t)(chunk + 1) * GAE_BLOCK + LOOKAHEAD >= n;  // ran off the end: x = 0 there
                if ((la.a == 0.0 && la.c == 0.0) || covers_all) {
                    carry_adv = la.b;
                    carry_ret = la.d;
                    done = true;
                }
            }
            // general path ahead: publish the aggregate first so that the chunk to the left can pass through this one
            // while we wait (only it can ask, and only if our prefix holds no trajectory end)
            if (!done && prefix_open && lane == 0) {
                put_double(rec + 0, TAG_AGG, agg.a);
                put_double(rec + 2, TAG_AGG, agg.b);
                put_double(rec + 4, TAG_AGG, agg.c);
                put_double(rec + 6, TAG_AGG, agg.d);
            }
            int j = chunk + 1;
            unsigned spins = 0;
            while (!done) {
                // lanes 0..11 read the 12 granules of chunk j (8 aggregate + 4 inclusive)
                u64 g = 0;
                if (lane < LB_AGG + LB_INC)
                    g = __hip_atomic_load(state + (size_t)j * LB_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned rtag = (unsigned)(g >> 32), val = (unsigned)g;  // (not `tag`: that is the launch's own, a kernel argument)
                const u64 inc_ok = __ballot(lane >= LB_AGG && lane < LB_AGG + LB_INC && rtag == TAG_INC);
                const u64 agg_ok = __ballot(lane < LB_AGG && rtag == TAG_AGG);
                auto dbl = [&](int first) {
                    const unsigned lo = __shfl(val, first), hi = __shfl(val, first + 1);
                    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
                };
                if (inc_ok == 0xF00ull) {
                    const double xa = dbl(8), xr = dbl(10);
                    carry_adv = acc.b + acc.a * xa;
                    carry_ret = acc.d + acc.c * xr;
                    done = true;
                } else if (agg_ok == 0xFFull) {
                    const Aff2 r = Aff2{dbl(0), dbl(2), dbl(4), dbl(6)};
                    acc = compose(acc, r);
                    if (acc.a == 0.0 && acc.c == 0.0) {  // a trajectory end cuts both recurrences: nothing further matters
                        carry_adv = acc.b;
                        carry_ret = acc.d;
                        done = true;
                    } else {
                        // an aggregate-only record means chunk j's own look-ahead failed, i.e. chunk j + 1's prefix holds no
                        // trajectory end, i.e. chunk j + 1 publishes too: the walk only ever visits publishers, and the
                        // rightmost chunk's carry is known (0), so it never publishes an aggregate and j never runs off the end
                        ++j;
                        spins = 0;
                    }
                } else {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > spin_limit) {
                        // The awaited record did not appear within the bound.  With per-launch tags and chunks taken in
                        // dispatch order (a chunk only ever waits on a workgroup dispatched before it) this cannot happen
                        // short of a defect, and then it must not pass silently: the carry is poisoned, so this chunk's
                        // outputs (and those of the chunks that chain through it) come out as NaN, and the workspace
                        // header counts the event.  (Recomputing the carry here from raw steps was tried: its registers
                        // pushed the common path into 152 bytes of scratch per lane, +29 % write traffic, +8 % time.)
                        if (lane == 0) atomicAdd(slow_word, 1u);
                        carry_adv = carry_ret = __longlong_as_double(0x7ff8000000000000ll);
                        done = true;
                    }
                }
            }
        }
        if (lane == 0) {
            if (prefix_open) {  // (a prefix with a trajectory end: the left neighbour's look-ahead never needs our records)
                put_double(rec + 8, TAG_INC, agg.b + agg.a * carry_adv);   // x at the first step of this chunk
                put_double(rec + 10, TAG_INC, agg.d + agg.c * carry_ret);
            }
            s_carry[0] = carry_adv;
            s_carry[1] = carry_ret;
        }
    }

}

// LOOP = false: one chunk per workgroup (the grid covers every chunk; the usual case) -- without the loop-carried state the
// kernel fits its 128-VGPR budget with no scratch at all.
#ifndef GAE_STORE_T_V
#define GAE_STORE_T_V 1
#endif
constexpr bool STORE_T = GAE_STORE_T_V;  // lane-contiguous output stores through an in-wave LDS transpose ([r4]; 0 = rounds 1-3's shape, for A/B builds)
template <bool LOOP>
__global__ __launch_bounds__(GAE_THREADS, 4) void gae_lookback_kernel(const float *__restrict__ rews,
                                                                    const float *__restrict__ dones,
                                                                    const float *__restrict__ trunc,
                                                                    const float *__restrict__ values, int64_t n,
                                                                    GaeParams p, u64 *__restrict__ state,
                                                                    const unsigned tag, const unsigned spin_limit,
                                                                    unsigned *__restrict__ slow_word, int n_blocks,
                                                                    float *__restrict__ vt_out,
                                                                    float *__restrict__ adv_out,
                                                                    float *__restrict__ ret_out) {
    __shared__ Aff2 wave_tot[4];
    __shared__ Aff2 wave_la[4];  // look-ahead window of the next chunk: per-wave composites (64 steps each)
    __shared__ double s_carry[2];
    constexpr int WAVE_SPAN = 64 * GAE_EPT;                                    // 512 consecutive steps per wave
    __shared__ __attribute__((aligned(16))) float o_lds[STORE_T ? 4 * 3 * WAVE_SPAN : 4];  // output transpose, 6 KiB per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Tags are unique per launch without a memset: `tag` is a kernel argument drawn from a process-wide host counter
    // (launch_gae), so every workgroup of a launch holds the same tag whatever its dispatch time, and records of earlier
    // launches (or never-written memory) carry a different tag in all 8 / 4 granules of a record.  (Round 1 kept the epoch
    // in the workspace header and let the owner of chunk 0 advance it on exit: a workgroup dispatched after that store --
    // possible when chunk 0's fast path needs nobody -- would have read the NEXT launch's tag and never matched its
    // neighbours.)  A captured launch would replay a frozen tag, so launch_gae takes the two-launch form under capture.
    const unsigned TAG_AGG = tag, TAG_INC = tag;
    // Chunks are taken right-to-left, grid-strided.  A chunk waits only on chunks to its right, i.e. on work of this
    // same round owned by lower block ids or on earlier rounds; the launcher sizes the grid to the number of
    // co-resident workgroups, so every awaited chunk belongs to a running workgroup whatever the dispatch order.
    for (int chunk = n_blocks - 1 - (int)blockIdx.x; chunk >= 0; chunk = LOOP ? chunk - (int)gridDim.x : -1) {
    const int64_t t0 = ((int64_t)chunk * GAE_THREADS + threadIdx.x) * GAE_EPT;

    // Look-ahead window = the first LOOKAHEAD (= 256) raw steps of the next chunk, ONE per thread, requested together
    // with the chunk's own steps: its memory round trip overlaps theirs instead of following the chunk scan, and the
    // window costs 5 registers per thread instead of 17 in wave 0 (which forced a second, serialised load wave there).
    const int64_t lt0 = (int64_t)(chunk + 1) * GAE_BLOCK + threadIdx.x;
    const bool l_ok = chunk + 1 < n_blocks && lt0 < n;
    float l_r = 0.f, l_d = 0.f, l_t = 0.f, l_v = 0.f, l_v1 = 0.f;
    if (l_ok) {
        l_r = rews[lt0];
        l_d = dones[lt0];
        l_t = trunc[lt0];
        l_v = values[lt0];
        l_v1 = values[lt0 + 1];
    }
    RawSteps raw;
    load_raw(rews, dones, trunc, values, t0, n, raw);
    {   // the window's composite while the chunk's own loads are still in flight (the window's were issued first)
        Aff2 l1 = aff_identity();  // steps past the end: identity (x = 0 there is handled by covers_all in resolve_carry)
        if (l_ok) l1 = step_affine(l_r, l_d, l_t, l_v, l_v1, p);
        const Aff2 ls = wave_suffix_scan(l1, lane);
        if (lane == 0) wave_la[wave] = ls;
    }
    __builtin_amdgcn_sched_barrier(0);
    Steps s;
    make_steps(raw, t0, n, p, s);
    const Aff2 mine = thread_composite(s, p);
    // Who will ever read this chunk's records?  Only the chunk to the left, and only if its look-ahead over OUR first
    // LOOKAHEAD steps finds no trajectory end (a chunk whose own carry came from the fast path publishes its inclusive value
    // alone, so nobody walks past it).  Threads 0 .. LOOKAHEAD / GAE_EPT - 1 of wave 0 hold those steps.
    const bool prefix_open = wave != 0 || (__ballot(lane < LOOKAHEAD / GAE_EPT && mine.a == 0.0 && mine.c == 0.0) == 0ull);
    // one shuffle scan serves both purposes: lane l gets the composite of lanes l..63 (needed for the outputs) and
    // lane 0's value is the wave total (needed for the chunk aggregate)
    const Aff2 inc = wave_suffix_scan(mine, lane);
    if (lane == 0) wave_tot[wave] = inc;
    __syncthreads();
    if (wave == 0) {  // (the aggregate and the records are wave 0's business alone)
        const Aff2 agg = compose(compose(wave_tot[0], wave_tot[1]), compose(wave_tot[2], wave_tot[3]));
        u64 *rec = state + (size_t)chunk * LB_STRIDE;
        resolve_carry(chunk, n_blocks, n, lane, wave, wave_la, agg, prefix_open, rec, state, TAG_AGG, TAG_INC, spin_limit, slow_word, s_carry);
    }

    // ---- exclusive suffix of this thread within the chunk, then the carry
    __syncthreads();  // publishes s_carry
    const double carry_adv = s_carry[0], carry_ret = s_carry[1];
    Aff2 after = shfl_down_aff(inc, 1);
    if (lane == 63) after = aff_identity();
    for (int w = wave + 1; w < 4; ++w) after = compose(after, wave_tot[w]);
    double x_adv = after.b + after.a * carry_adv;
    double x_ret = after.d + after.c * carry_ret;

    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t wbase = ((int64_t)chunk * GAE_THREADS + wave * 64) * GAE_EPT;  // first step of this wave
    const bool full_wave = STORE_T && wbase + WAVE_SPAN <= n;                     // (wave-uniform)
    // [r4] The outputs leave LANE-CONTIGUOUS.  As the scan holds them a wave-instruction would store 64 x 16 bytes at a 32-byte
    // stride -- half of each of sixteen 128-byte lines, the other halves one instruction later -- and on data that streams from
    // and to HBM (not the Infinity-Cache-resident re-scan rounds 1-3 timed) it is the STORE shape that costs: same seven streams,
    // no scan, 8-consecutive loads: 14.9 us with 8-consecutive stores, 13.4 us with lane-contiguous ones; the load shape changes
    // nothing (14.9 us; profiles/r04_gae_floor.txt).  Each wave transposes its 512 steps per output through 2 KiB of its own LDS
    // (in-wave: the LDS operations of a wave execute in order, no workgroup barrier) and stores 1 KiB of contiguous bytes per
    // instruction.  A group of four steps goes to LDS as soon as it is formed (12 live output registers, not 24).
    float *o = o_lds + (STORE_T ? wave * (3 * WAVE_SPAN) : 0);
#pragma unroll
    for (int h = GAE_EPT / 4 - 1; h >= 0; --h) {
        f4 q_adv, q_vt, q_ret;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int e = 4 * h + j;
            x_adv = s.b_adv[e] + coef_adv(s, e, p) * x_adv;
            x_ret = (double)s.r[e] + coef_ret(s, e, p) * x_ret;
            q_adv[j] = (float)x_adv;
            q_vt[j] = (float)((double)s.v[e] + x_adv);
            q_ret[j] = (float)x_ret;
        }
        if (full_wave) {
            *reinterpret_cast<f4 *>(o + lane * GAE_EPT + 4 * h) = q_adv;
            *reinterpret_cast<f4 *>(o + WAVE_SPAN + lane * GAE_EPT + 4 * h) = q_vt;
            *reinterpret_cast<f4 *>(o + 2 * WAVE_SPAN + lane * GAE_EPT + 4 * h) = q_ret;
        } else if (t0 + GAE_EPT <= n) {
            *reinterpret_cast<f4 *>(adv_out + t0 + 4 * h) = q_adv;
            *reinterpret_cast<f4 *>(vt_out + t0 + 4 * h) = q_vt;
            *reinterpret_cast<f4 *>(ret_out + t0 + 4 * h) = q_ret;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t0 + 4 * h + j < n) {
                    adv_out[t0 + 4 * h + j] = q_adv[j];
                    vt_out[t0 + 4 * h + j] = q_vt[j];
                    ret_out[t0 + 4 * h + j] = q_ret[j];
                }
        }
    }
    if (full_wave) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h = 0; h < GAE_EPT / 4; ++h) {
            const int k = h * 256 + lane * 4;
            *reinterpret_cast<f4 *>(adv_out + wbase + k) = *reinterpret_cast<const f4 *>(o + k);
            *reinterpret_cast<f4 *>(vt_out + wbase + k) = *reinterpret_cast<const f4 *>(o + WAVE_SPAN + k);
            *reinterpret_cast<f4 *>(ret_out + wbase + k) = *reinterpret_cast<const f4 *>(o + 2 * WAVE_SPAN + k);
        }
    }
    __syncthreads();  // LDS (lds4, wave_tot, s_carry, o_lds) is reused by the next round
    }
}

static int g_gae_oversubscribe = 0;  // rlppo_dbg_set(22, 0/1)
void set_gae_oversubscribe(int v) { g_gae_oversubscribe = v; }
static int g_gae_algo = 1;  // 1 = single-pass look-back (default), 0 = two launches (summary + apply)
void set_gae_algo(int a) { g_gae_algo = a; }
static unsigned g_gae_spin_limit = LB_SPIN_LIMIT;  // rlppo_dbg_set(21, v): tests set 0 to force the timeout path
void set_gae_spin_limit(int v) { g_gae_spin_limit = v < 0 ? LB_SPIN_LIMIT : (unsigned)v; }

// workspace: [0,16) header: word 1 = number of look-back waits that timed out (their chunks' outputs are NaN; never reset by
// the library) | look-back state (128 B per chunk) ; the two-launch path uses the same
// region for its per-chunk composites (32 B per chunk)
size_t gae_workspace_bytes(int64_t n) { return 16 + (size_t)(cdiv(n > 0 ? n : 1, GAE_BLOCK)) * LB_STRIDE * sizeof(u64); }

int launch_gae(hipStream_t st, const float *rews, const float *dones, const float *trunc, const float *values, int64_t n,
               double gamma, double lmbda, float ret_std, float *vt, float *adv, float *ret, void *ws, size_t ws_bytes) {
    if (n == 0) return 0;
    RLPPO_CHECK_ARG(n > 0, "gae: n=%ld", (long)n);
    RLPPO_CHECK_ARG(((uintptr_t)rews | (uintptr_t)dones | (uintptr_t)trunc | (uintptr_t)values | (uintptr_t)vt |
                     (uintptr_t)adv | (uintptr_t)ret | (uintptr_t)ws) % 16 == 0,
                    "gae: arrays and workspace must be 16-byte aligned");
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
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st) RLPPO_HIP(hipStreamIsCapturing(st, &cap));
    if (g_gae_algo >= 1 && cap == hipStreamCaptureStatusNone) {
        unsigned *hdr = reinterpret_cast<unsigned *>(ws);
        u64 *state = reinterpret_cast<u64 *>(reinterpret_cast<char *>(ws) + 16);
        // co-resident workgroups of this kernel, per device (queried once for each device id: a process may drive several).
        // The LOOP form (more chunks than the grid) makes a workgroup of round k wait on one of round k - 1, which is only safe
        // while the whole grid is resident -- and other streams' kernels may hold some of the slots the occupancy query counts:
        // the grid is therefore kept one workgroup per CU BELOW the theoretical occupancy.
        static std::atomic<int> resident_by_dev[64], cus_by_dev[64];
        int dev = 0;
        RLPPO_HIP(hipGetDevice(&dev));
        RLPPO_CHECK_ARG(dev >= 0 && dev < 64, "gae: device id %d", dev);
        int resident = resident_by_dev[dev].load(std::memory_order_acquire);
        if (resident == 0) {
            int per_cu = 0;
            hipDeviceProp_t prop;
            RLPPO_HIP(hipGetDeviceProperties(&prop, dev));
            RLPPO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gae_lookback_kernel<true>, GAE_THREADS, 0));
            // the occupancy API over-reports by one block per CU only in the SGPR-limited 7-8 blocks/CU regime
            // (MI355X_MICROARCH.md "Residency and cooperative launch"); this kernel is VGPR-limited far below that
            per_cu = per_cu > 6 ? 6 : (per_cu < 1 ? 1 : per_cu);
            resident = per_cu * prop.multiProcessorCount;
            cus_by_dev[dev].store(prop.multiProcessorCount, std::memory_order_relaxed);
            resident_by_dev[dev].store(resident, std::memory_order_release);
        }
        const int cus = cus_by_dev[dev].load(std::memory_order_relaxed);
        const int loop_grid = resident - cus > 0 ? resident - cus : resident;  // LOOP form: one workgroup per CU of margin
        // g_gae_oversubscribe: one workgroup per chunk even beyond the resident capacity.  Chunks are taken right to left in
        // workgroup-id order and every XCD dispatches its workgroups in id order, so the smallest unfinished chunk always holds
        // (or is next in line for) a slot and waits on nothing unfinished: no deadlock without co-residency, and the loads of a
        // later wave of workgroups overlap the stores of an earlier one instead of all workgroups moving in lockstep.
        // Up to the theoretical occupancy every chunk gets its own workgroup (the loop-free kernel: safe whatever is resident,
        // by the dispatch-order argument above); beyond it the LOOP form runs on the margin-reduced grid.
        int grid = (nb <= resident || g_gae_oversubscribe) ? nb : loop_grid;
        // per-launch tag: process-wide counter seeded from the clock (a recycled workspace may hold records of another
        // process's launches), never 0 (zero-filled memory)
        static std::atomic<unsigned> g_tag{0};
        if (g_tag.load(std::memory_order_relaxed) == 0) {
            unsigned expect = 0;
            const unsigned seed = (unsigned)std::chrono::steady_clock::now().time_since_epoch().count() | 1u;
            g_tag.compare_exchange_strong(expect, seed);
        }
        unsigned tag = g_tag.fetch_add(1, std::memory_order_relaxed) + 1;
        if (tag == 0) tag = g_tag.fetch_add(1, std::memory_order_relaxed) + 1;
        auto *kern = grid == nb ? gae_lookback_kernel<false> : gae_lookback_kernel<true>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(GAE_THREADS), 0, st, rews, dones, trunc, values, n, p, state, tag, g_gae_spin_limit,
                           hdr + 1, nb, vt, adv, ret);
        RLPPO_LAUNCH_CHECK();
        return 0;
    }
    Aff2 *summ = reinterpret_cast<Aff2 *>(reinterpret_cast<char *>(ws) + 16);
    hipLaunchKernelGGL(gae_summary_kernel, dim3(nb), dim3(GAE_THREADS), 0, st, rews, dones, trunc, values, n, p, summ);
    RLPPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(gae_apply_kernel, dim3(nb), dim3(GAE_THREADS), 0, st, rews, dones, trunc, values, n, p, summ, nb,
                       vt, adv, ret);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo
