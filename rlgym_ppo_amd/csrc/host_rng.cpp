// host_rng.cpp -- HOST code: numpy's legacy RandomState stream (MT19937, init_genrand seeding, 32-bit masked
// rejection sampling, reverse Fisher-Yates), i.e. the index stream of ExperienceBuffer.get_all_batches_shuffled
// (reference: rlgym_ppo/ppo/experience_buffer.py:52,97-98 -> numpy.random.RandomState(seed).permutation(n)).
// numpy is a third-party dependency of the reference (requirements.txt:7); the algorithm restated here is the
// published one of numpy/random/src/mt19937/mt19937.c and legacy-distributions.c (random_interval) and
// mtrand.pyx (_shuffle_raw), pinned by tests against numpy itself and by fixture tests/golden/g6_shuffle.npz.
// It exists so that the permutation of a 512k-sample buffer takes ~3 ms instead of numpy's ~12 ms and can run
// outside the GIL while the GPU works on the previous epoch.
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/rlppo.h"

namespace {
constexpr int N = 624, M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

inline uint32_t twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & UPPER) | (nxt & LOWER);
    return far ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}

// One 624-word regeneration, eight words per step: word kk needs the OLD kk+1 (loaded before kk..kk+7 are stored) and
// either the old kk+397 (first 227 words) or the NEW kk-227 (written at least 220 words earlier).
inline void regen(uint32_t *mt) {
    const __m256i upper = _mm256_set1_epi32((int)UPPER), lower = _mm256_set1_epi32((int)LOWER);
    const __m256i mat = _mm256_set1_epi32((int)MATRIX_A), one = _mm256_set1_epi32(1);
    auto twist8 = [&](int kk, int far) {
        const __m256i cur = _mm256_loadu_si256((const __m256i *)(mt + kk)), nxt = _mm256_loadu_si256((const __m256i *)(mt + kk + 1));
        const __m256i y = _mm256_or_si256(_mm256_and_si256(cur, upper), _mm256_and_si256(nxt, lower));
        const __m256i odd = _mm256_sub_epi32(_mm256_setzero_si256(), _mm256_and_si256(y, one));  // 0 or ~0
        const __m256i r = _mm256_xor_si256(_mm256_xor_si256(_mm256_loadu_si256((const __m256i *)(mt + far)), _mm256_srli_epi32(y, 1)),
                                           _mm256_and_si256(odd, mat));
        _mm256_storeu_si256((__m256i *)(mt + kk), r);
    };
    int kk = 0;
    for (; kk + 8 <= N - M; kk += 8) twist8(kk, kk + M);
    for (; kk < N - M; kk++) mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + M]);
    for (; kk + 8 <= N - 1; kk += 8) twist8(kk, kk + (M - N));
    for (; kk < N - 1; kk++) mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + (M - N)]);
    mt[N - 1] = twist(mt[N - 1], mt[0], mt[M - 1]);
}

inline uint32_t next32(uint32_t *st) {
    uint32_t &pos = st[N];
    if (pos >= (uint32_t)N) {
        regen(st);
        pos = 0;
    }
    uint32_t y = st[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// permutevar8x32 index vectors that move the lanes selected by an 8-bit mask to the front, in order
struct CompressLut {
    alignas(32) uint32_t idx[256][8];
    CompressLut() {
        for (int m = 0; m < 256; m++) {
            int c = 0;
            for (int k = 0; k < 8; k++)
                if (m >> k & 1) idx[m][c++] = (uint32_t)k;
            for (; c < 8; c++) idx[m][c] = 0;
        }
    }
};
inline const CompressLut &compress_lut() {
    static const CompressLut lut;
    return lut;
}
}  // namespace

extern "C" {

int rlppo_mt19937_seed(uint32_t *st, uint32_t seed) {
    if (!st) return RLPPO_ERR_ARG;
    st[0] = seed;
    for (int i = 1; i < N; i++) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
    st[N] = N;
    return 0;
}

int rlppo_mt19937_permutation(uint32_t *st, int64_t n, int64_t *out) {
    if (!st || n < 0 || (n > 0 && !out) || n > 0x7fffffffLL) return RLPPO_ERR_ARG;
    // Same draws and swaps as numpy's _shuffle_raw, restructured for the CPU: (1) the generator is run a whole
    // 624-word block at a time (regenerate + temper into a cache); (2) the rejection test of random_interval is
    // branch-free (every stream word is consumed exactly once; an accepted word advances i); (3) the random targets
    // of a batch of consecutive i are drawn first and their cache lines prefetched, then the swaps are applied in
    // order on a 32-bit working copy (2 MB at 512k samples).
    static thread_local std::vector<int32_t> work;
    work.resize((size_t)n);
    int32_t *w = work.data();
    for (int64_t i = 0; i < n; i++) w[i] = (int32_t)i;

    uint32_t cache[N];
    uint32_t pos = st[N] > (uint32_t)N ? (uint32_t)N : st[N];
    auto temper_block = [&](uint32_t from) {
        for (uint32_t k = from; k < (uint32_t)N; k++) {
            uint32_t y = st[k];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            cache[k] = y;
        }
    };
    temper_block(pos);

    constexpr int BATCH = 64;
    uint32_t js[BATCH + 1];
    int64_t i = n - 1;
    while (i >= 1) {
        const int want = (int)(i < BATCH ? i : BATCH);
        int cnt = 0;
        uint32_t ii = (uint32_t)i;
        while (cnt < want) {
            // ii stays in (lo, mask] for a long run, so the mask is hoisted out of the accept chain (cmp + sub only)
            const uint32_t mask = 0xffffffffu >> __builtin_clz(ii);  // smallest all-ones mask >= ii
            const uint32_t lo = mask >> 1;
            while (cnt < want && ii > lo) {
                if (pos >= (uint32_t)N) {
                    regen(st);
                    temper_block(0);
                    pos = 0;
                }
                const uint32_t v = cache[pos++] & mask;
                const uint32_t acc = v <= ii;
                js[cnt] = v;
                cnt += (int)acc;
                ii -= acc;
            }
        }
        for (int k = 0; k < want; k++) __builtin_prefetch(&w[js[k]], 1, 1);
        for (int k = 0; k < want; k++) {
            const int64_t t_i = i - k;
            const int32_t t = w[js[k]];
            w[js[k]] = w[t_i];
            w[t_i] = t;
        }
        i -= want;
    }
    st[N] = pos;
    for (int64_t k = 0; k < n; k++) out[k] = w[k];
    return 0;
}

// ---- the same permutation in two phases --------------------------------------------------------------------------
// The swap targets j_i of the reverse Fisher-Yates loop depend on the generator only, not on the array being shuffled.
// Phase 1 (rlppo_mt19937_draw_targets) is the inherently serial part: it consumes the stream exactly as
// rlppo_mt19937_permutation does and records targets[t] = j_i for i = n-1-t (draw order).  Phase 2
// (rlppo_apply_swap_targets) applies the swaps; it touches no generator state, so the swaps of epoch e can run on another
// thread while phase 1 already draws epoch e+1 (engine.ShufflePipeline).  With 8 data-parallel ranks the GPU share of an
// epoch (~1.1 ms) is shorter than the fused permutation (~3 ms); the serial critical path is then phase 1 alone.
int rlppo_mt19937_draw_targets(uint32_t *st, int64_t n, uint32_t *targets) {
    if (!st || n < 0 || (n > 1 && !targets) || n > 0x7fffffffLL) return RLPPO_ERR_ARG;
    if (n < 2) return 0;
    alignas(32) uint32_t cache[N + 8];
    uint32_t pos = st[N] > (uint32_t)N ? (uint32_t)N : st[N];
    auto temper_block = [&](uint32_t from) {
        for (uint32_t k = from; k < (uint32_t)N; k++) {
            uint32_t y = st[k];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            cache[k] = y;
        }
    };
    temper_block(pos);
    const CompressLut &lut = compress_lut();
    uint32_t *out = targets;
    uint32_t ii = (uint32_t)(n - 1);
    while (ii >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz(ii);  // smallest all-ones mask >= ii
        const uint32_t lo = mask >> 1;                             // the run ends when ii drops to lo
        const __m256i vmask = _mm256_set1_epi32((int)mask);
        while (ii > lo) {
            if (pos >= (uint32_t)N) {
                regen(st);
                temper_block(0);
                pos = 0;
            }
            // Eight stream words at a time.  Word k of a group is accepted iff (w & mask) <= ii - (accepted before it), so
            // (w & mask) <= ii - 8 is accepted and (w & mask) > ii is rejected whatever the words before it did; a group in
            // which every word is one or the other (all but ~64/ii of them) is compacted with one permute and stored in
            // draw order (the 8 written slots always lie inside the caller's n entries: at least 8 targets remain).  masks
            // are < 2^31 (n <= INT32_MAX), so signed compares are exact.
            while (pos + 8 <= (uint32_t)N && ii > lo + 8) {
                const __m256i w = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(cache + pos)), vmask);
                const __m256i rej = _mm256_cmpgt_epi32(w, _mm256_set1_epi32((int)ii));
                const __m256i acc = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)(ii - 8 + 1)), w);
                const int macc = _mm256_movemask_ps(_mm256_castsi256_ps(acc));
                const int mrej = _mm256_movemask_ps(_mm256_castsi256_ps(rej));
                if ((macc | mrej) != 0xff) break;  // an undecided word: this group goes through the scalar loop
                _mm256_storeu_si256((__m256i *)out, _mm256_permutevar8x32_epi32(w, _mm256_load_si256((const __m256i *)lut.idx[macc])));
                const uint32_t c = (uint32_t)__builtin_popcount((unsigned)macc);
                out += c;
                ii -= c;
                pos += 8;
            }
            for (int k = 0; k < 8 && pos < (uint32_t)N && ii > lo; k++) {
                const uint32_t v = cache[pos++] & mask;
                const uint32_t a = v <= ii;
                *out = v;
                out += a;
                ii -= a;
            }
        }
    }
    st[N] = pos;
    return 0;
}

int rlppo_apply_swap_targets(int64_t n, const uint32_t *targets, int64_t *out) {
    if (n < 0 || (n > 0 && !out) || (n > 1 && !targets) || n > 0x7fffffffLL) return RLPPO_ERR_ARG;
    static thread_local std::vector<int32_t> work;
    work.resize((size_t)n);
    int32_t *w = work.data();
    for (int64_t i = 0; i < n; i++) w[i] = (int32_t)i;
    constexpr int64_t AHEAD = 32;  // the targets are known in advance: their cache lines are requested 32 swaps early
    for (int64_t t = 0; t + 1 < n; t++) {
        if (t + AHEAD < n - 1) __builtin_prefetch(&w[targets[t + AHEAD]], 1, 1);
        const int64_t i = n - 1 - t;
        const uint32_t j = targets[t];
        if ((int64_t)j > i) return RLPPO_ERR_ARG;  // not a target vector of rlppo_mt19937_draw_targets
        const int32_t v = w[j];
        w[j] = w[i];
        w[i] = v;
    }
    for (int64_t k = 0; k < n; k++) out[k] = w[k];
    return 0;
}
}

// ------------------------------------------------------------------------------------------------------------------------
// torch.empty(n).exponential_(lambda) of PyTorch's CPU generator, bit for bit -- the noise torch.multinomial(probs, 1, True)
// and Categorical.sample() draw in the reference's rollout (rlgym_ppo/ppo/discrete_policy.py:59, util/torch_functions.py:115;
// SURVEY.md section 8(a1)).  PyTorch is a third-party dependency of the reference (requirements.txt, unpinned `torch>1.13`);
// the algorithm restated here is the published one of its CPU path (ATen/native/cpu/DistributionTemplates.h
// exponential_kernel_default -> ATen/core/DistributionsHelper.h + TransformationHelper.h, c10 MT19937RNGEngine.h), pinned by
// tests/test_abi_and_layout.py against torch itself on every box:
//   per element: r64 = (engine() << 32) | engine()            two tempered MT19937 words, high word first (random64)
//                u   = (r64 & (2^53 - 1)) * 2^-53              uniform_real_distribution<double>
//                q   = (float)(-1/lambda * log1p(-u))          transformation::exponential, CPU branch, double math
// The stream phase (MT19937 regeneration + tempering) is serial and vectorised; the transform is embarrassingly parallel
// and runs on `threads` std::threads.  torch's own kernel is serial at ~12-26 ns per element (4.7 ms for the 4096 x 90
// draw of one rollout step, the whole cost of the bit-exact rollout mode in round 1).
//
// `state` is the generator's serialised state as torch.get_rng_state() returns it (CPUGeneratorImpl.cpp,
// CPUGeneratorImplState: 5056 bytes): { u64 seed; i32 left; i32 seeded; u64 next; u64 mt[624]; double normal_x, normal_y,
// normal_rho; i32 normal_is_valid; float next_float_normal_sample; bool valid }.  It is advanced in place exactly as the
// generator would advance (left/next follow c10::mt19937: `if (--left == 0) next_state(); y = state[next++]`).
#include <math.h>

#include <thread>

namespace {
struct TorchState {
    uint64_t seed;
    int32_t left, seeded;
    uint64_t next;
    uint64_t mt[N];
};
static_assert(sizeof(TorchState) == 8 + 8 + 8 + 8 * N, "layout of the legacy generator state header");

// The stream phase 16 words per step where the host has AVX-512 (checked at run time): one 624-word regeneration and the tempering
// of `take` words from src to dst.  Same recurrences as regen() / temper8(); the 16-word steps stay inside the 227-word distance
// between a word and the newest word it depends on.
__attribute__((target("avx512f"))) static inline void twist16(uint32_t *mt, int kk, int far) {
    const __m512i upper = _mm512_set1_epi32((int)UPPER), lower = _mm512_set1_epi32((int)LOWER);
    const __m512i mat = _mm512_set1_epi32((int)MATRIX_A), one = _mm512_set1_epi32(1);
    const __m512i cur = _mm512_loadu_si512((const void *)(mt + kk)), nxt = _mm512_loadu_si512((const void *)(mt + kk + 1));
    const __m512i y = _mm512_or_si512(_mm512_and_si512(cur, upper), _mm512_and_si512(nxt, lower));
    const __m512i odd = _mm512_sub_epi32(_mm512_setzero_si512(), _mm512_and_si512(y, one));  // 0 or ~0
    const __m512i r = _mm512_xor_si512(_mm512_xor_si512(_mm512_loadu_si512((const void *)(mt + far)), _mm512_srli_epi32(y, 1)),
                                       _mm512_and_si512(odd, mat));
    _mm512_storeu_si512((void *)(mt + kk), r);
}
__attribute__((target("avx512f"))) static void regen512(uint32_t *mt) {
    int kk = 0;
    for (; kk + 16 <= N - M; kk += 16) twist16(mt, kk, kk + M);
    for (; kk < N - M; kk++) mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + M]);
    for (; kk + 16 <= N - 1; kk += 16) twist16(mt, kk, kk + (M - N));
    for (; kk < N - 1; kk++) mt[kk] = twist(mt[kk], mt[kk + 1], mt[kk + (M - N)]);
    mt[N - 1] = twist(mt[N - 1], mt[0], mt[M - 1]);
}
__attribute__((target("avx512f"))) static int temper512(const uint32_t *src, uint32_t *dst, int take) {
    int k = 0;
    for (; k + 16 <= take; k += 16) {
        __m512i y = _mm512_loadu_si512((const void *)(src + k));
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), _mm512_set1_epi32((int)0x9d2c5680u)));
        y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), _mm512_set1_epi32((int)0xefc60000u)));
        y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
        _mm512_storeu_si512((void *)(dst + k), y);
    }
    return k;
}
inline void temper8(const uint32_t *src, uint32_t *dst) {
    __m256i y = _mm256_loadu_si256((const __m256i *)src);
    y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
    y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), _mm256_set1_epi32((int)0x9d2c5680u)));
    y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), _mm256_set1_epi32((int)0xefc60000u)));
    y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
    _mm256_storeu_si256((__m256i *)dst, y);
}
}  // namespace

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

namespace {
// ---- the transform out = (float)(scale * log1p(-u)), u = k 2^-53, certified against libm
// libm's log1p costs ~7 ns per element (2.6 ms for the 4096 x 90 draw of one rollout step); torch's result is what it returns,
// rounded to float32.  A vectorised logarithm cannot be trusted to reproduce libm's last bit -- but only the FLOAT32 rounding of
// the product is needed, and that is decided by any double within 1e-13 relative of libm's unless the value sits that close to a
// float32 rounding boundary (3 elements per million).  So: x = 1 - u is exact in double (u is a multiple of 2^-53), log(x) comes
// from an AVX2 evaluation -- x = 2^e m, m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + ... +
// s^22/23), e ln2 added as hi + lo parts: relative error below 2e-15 -- and an element is accepted only if y (1 - 1e-13) and
// y (1 + 1e-13) round to the SAME float32; libm's value (within 1 ulp of the truth, times one rounded product) lies between
// them, so it rounds to that float as well.  Everything else (and u = 0, whose result is a signed zero) goes through libm.
// The tests pin the outcome against torch itself and the two paths against each other over 2^24 values.
static int g_exp_fast = 1;  // rlppo_dbg_set(24, .)
inline void exp_exact(const uint32_t *w, float *out, int64_t i, double scale) {
    const uint64_t r64 = ((uint64_t)w[2 * i] << 32) | (uint64_t)w[2 * i + 1];
    const double u = (double)(r64 & ((1ULL << 53) - 1)) * (1.0 / 9007199254740992.0);
    out[i] = (float)(scale * log1p(-u));
}
// the same evaluation 8 elements at a time where the host has AVX-512 (checked at run time; the file is built for AVX2)
__attribute__((target("avx512f,avx512dq"))) static int64_t exp_transform_512(const uint32_t *w, float *out, int64_t i, int64_t hi,
                                                                             double scale) {
    const __m512d one = _mm512_set1_pd(1.0), half = _mm512_set1_pd(0.5), sqrt2 = _mm512_set1_pd(1.4142135623730951);
    const __m512d inv53 = _mm512_set1_pd(1.0 / 9007199254740992.0);
    const __m512d ln2_hi = _mm512_set1_pd(6.93147180369123816490e-01), ln2_lo = _mm512_set1_pd(1.90821492927058770002e-10);
    const __m512d vscale = _mm512_set1_pd(scale), eps_lo = _mm512_set1_pd(1.0 - 1e-13), eps_hi = _mm512_set1_pd(1.0 + 1e-13);
    const __m512i mask53 = _mm512_set1_epi64((1LL << 53) - 1), mant = _mm512_set1_epi64(0x000fffffffffffffLL);
    const __m512i expo1 = _mm512_set1_epi64(0x3ff0000000000000LL), bias = _mm512_set1_epi64(1023), ione = _mm512_set1_epi64(1);
    for (; i + 8 <= hi; i += 8) {
        const __m512i r64 = _mm512_shuffle_epi32(_mm512_loadu_si512((const void *)(w + 2 * i)), (_MM_PERM_ENUM)0xB1);
        const __m512d kd = _mm512_cvtepi64_pd(_mm512_and_si512(r64, mask53));       // exact: < 2^53
        const __m512d x = _mm512_sub_pd(one, _mm512_mul_pd(kd, inv53));             // exact
        const __m512i xb = _mm512_castpd_si512(x);
        __m512i e = _mm512_sub_epi64(_mm512_srli_epi64(xb, 52), bias);
        __m512d m = _mm512_castsi512_pd(_mm512_or_si512(_mm512_and_si512(xb, mant), expo1));
        const __mmask8 big = _mm512_cmp_pd_mask(m, sqrt2, _CMP_GT_OQ);
        m = _mm512_mask_mul_pd(m, big, m, half);
        e = _mm512_mask_add_epi64(e, big, e, ione);
        const __m512d ed = _mm512_cvtepi64_pd(e);
        const __m512d sq = _mm512_div_pd(_mm512_sub_pd(m, one), _mm512_add_pd(m, one)), s2 = _mm512_mul_pd(sq, sq);
        __m512d p = _mm512_set1_pd(1.0 / 23.0);
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 21.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 19.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 17.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 15.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 13.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 11.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 9.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 7.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 5.0));
        p = _mm512_fmadd_pd(p, s2, _mm512_set1_pd(1.0 / 3.0));
        p = _mm512_fmadd_pd(p, s2, one);
        const __m512d logm = _mm512_mul_pd(_mm512_add_pd(sq, sq), p);
        const __m512d lg = _mm512_fmadd_pd(ed, ln2_hi, _mm512_fmadd_pd(ed, ln2_lo, logm));
        const __m512d y = _mm512_mul_pd(vscale, lg);
        const __m256 f0 = _mm512_cvtpd_ps(_mm512_mul_pd(y, eps_lo)), f1 = _mm512_cvtpd_ps(_mm512_mul_pd(y, eps_hi));
        const int same = _mm256_movemask_ps(_mm256_cmp_ps(f0, f1, _CMP_EQ_OQ));
        const int nonzero = (int)_mm512_cmp_pd_mask(lg, _mm512_setzero_pd(), _CMP_NEQ_OQ);
        _mm256_storeu_ps(out + i, f0);
        const int bad = ~(same & nonzero) & 255;
        if (bad)
            for (int t = 0; t < 8; ++t)
                if ((bad >> t) & 1) exp_exact(w, out, i + t, scale);
    }
    return i;
}
inline void exp_transform(const uint32_t *w, float *out, int64_t lo, int64_t hi, double scale) {
    auto exact = [&](int64_t i) { exp_exact(w, out, i, scale); };
    int64_t i = lo;
    static const bool have512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    if (g_exp_fast && have512) i = exp_transform_512(w, out, i, hi, scale);
    if (g_exp_fast) {
        const __m256d one = _mm256_set1_pd(1.0), half = _mm256_set1_pd(0.5), sqrt2 = _mm256_set1_pd(1.4142135623730951);
        const __m256d two52 = _mm256_set1_pd(4503599627370496.0);
        const __m256d two84_52 = _mm256_set1_pd(19342813118337666422669312.0);  // 2^84 + 2^52
        const __m256d inv53 = _mm256_set1_pd(1.0 / 9007199254740992.0);
        const __m256d ln2_hi = _mm256_set1_pd(6.93147180369123816490e-01), ln2_lo = _mm256_set1_pd(1.90821492927058770002e-10);
        const __m256d vscale = _mm256_set1_pd(scale), eps_lo = _mm256_set1_pd(1.0 - 1e-13), eps_hi = _mm256_set1_pd(1.0 + 1e-13);
        const __m256i mask53 = _mm256_set1_epi64x((1LL << 53) - 1), lo32 = _mm256_set1_epi64x(0xffffffffLL);
        const __m256i mant = _mm256_set1_epi64x(0x000fffffffffffffLL), expo1 = _mm256_set1_epi64x(0x3ff0000000000000LL);
        const __m256i magic_lo = _mm256_set1_epi64x(0x4330000000000000LL), magic_hi = _mm256_set1_epi64x(0x4530000000000000LL);
        const __m256i bias = _mm256_set1_epi64x(1023);
        for (; i + 4 <= hi; i += 4) {
            // r64 = (w[2i] << 32) | w[2i+1]: the pairs are stored high word first, i.e. as 64-bit lanes with the halves swapped
            const __m256i raw = _mm256_loadu_si256((const __m256i *)(w + 2 * i));
            const __m256i r64 = _mm256_shuffle_epi32(raw, 0xB1);
            const __m256i k = _mm256_and_si256(r64, mask53);
            // exact int64 (< 2^53) -> double: low 32 bits and high 21 bits through the 2^52 / 2^84 exponent tricks
            const __m256d dlo = _mm256_castsi256_pd(_mm256_or_si256(_mm256_and_si256(k, lo32), magic_lo));
            const __m256d dhi = _mm256_castsi256_pd(_mm256_or_si256(_mm256_srli_epi64(k, 32), magic_hi));
            const __m256d kd = _mm256_add_pd(_mm256_sub_pd(dhi, two84_52), dlo);  // (hi 2^32 + 2^84 - 2^84 - 2^52) + (lo + 2^52): exact
            const __m256d x = _mm256_sub_pd(one, _mm256_mul_pd(kd, inv53));       // exact: a multiple of 2^-53 in (0, 1]
            const __m256i xb = _mm256_castpd_si256(x);
            __m256i e = _mm256_sub_epi64(_mm256_srli_epi64(xb, 52), bias);        // x > 0: no sign bit
            __m256d m = _mm256_castsi256_pd(_mm256_or_si256(_mm256_and_si256(xb, mant), expo1));  // [1, 2)
            const __m256d big = _mm256_cmp_pd(m, sqrt2, _CMP_GT_OQ);
            m = _mm256_blendv_pd(m, _mm256_mul_pd(m, half), big);
            e = _mm256_sub_epi64(e, _mm256_castpd_si256(big));                    // big lanes are all-ones = -1: e += 1
            // e in [-53, 0] -> double through the same 2^52 trick on e + 2^52's mantissa (e + 64 is a small non-negative integer)
            const __m256d ed = _mm256_sub_pd(_mm256_castsi256_pd(_mm256_or_si256(_mm256_add_epi64(e, _mm256_set1_epi64x(64)), magic_lo)),
                                             _mm256_add_pd(two52, _mm256_set1_pd(64.0)));
            const __m256d sn = _mm256_sub_pd(m, one), sd = _mm256_add_pd(m, one);
            const __m256d sq = _mm256_div_pd(sn, sd), s2 = _mm256_mul_pd(sq, sq);
            __m256d p = _mm256_set1_pd(1.0 / 23.0);
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 21.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 19.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 17.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 15.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 13.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 11.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 9.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 7.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 5.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), _mm256_set1_pd(1.0 / 3.0));
            p = _mm256_add_pd(_mm256_mul_pd(p, s2), one);
            const __m256d logm = _mm256_mul_pd(_mm256_add_pd(sq, sq), p);
            const __m256d lg = _mm256_add_pd(_mm256_mul_pd(ed, ln2_hi), _mm256_add_pd(logm, _mm256_mul_pd(ed, ln2_lo)));
            const __m256d y = _mm256_mul_pd(vscale, lg);
            const __m128 f0 = _mm256_cvtpd_ps(_mm256_mul_pd(y, eps_lo)), f1 = _mm256_cvtpd_ps(_mm256_mul_pd(y, eps_hi));
            const int same = _mm_movemask_ps(_mm_cmpeq_ps(f0, f1));
            const int nonzero = _mm256_movemask_pd(_mm256_cmp_pd(lg, _mm256_setzero_pd(), _CMP_NEQ_OQ));
            _mm_storeu_ps(out + i, f0);
            const int bad = ~(same & nonzero) & 15;
            if (bad)
                for (int t = 0; t < 4; ++t)
                    if ((bad >> t) & 1) exact(i + t);
        }
    }
    for (; i < hi; i++) exact(i);
}

struct ExpJob {
    static constexpr int64_t SLICE = 2048;
    const uint32_t *w = nullptr;
    float *out = nullptr;
    int64_t n = 0;
    double scale = -1.0;
    std::atomic<int64_t> published{0};  // elements whose two words are in `w`
    std::atomic<int64_t> claimed{0};    // next element nobody has taken yet
    std::atomic<int> busy{0};           // helper threads still inside work()
    void work() {
        for (;;) {
            const int64_t lo = claimed.fetch_add(SLICE, std::memory_order_relaxed);
            if (lo >= n) return;
            const int64_t hi = lo + SLICE < n ? lo + SLICE : n;
            while (published.load(std::memory_order_acquire) < hi) _mm_pause();  // the stream phase is ~3x faster than one transformer
            exp_transform(w, out, lo, hi, scale);
        }
    }
};

// Persistent helper threads (created on first use, detached: they sleep on a condition variable between draws and die with
// the process).  One job at a time; callers from different threads serialise on `gate`.
struct ExpPool {
    std::mutex m, gate;
    std::condition_variable cv;
    ExpJob *job = nullptr;
    uint64_t generation = 0;
    int wanted = 0, spawned = 0;
    static ExpPool &get() {
        static ExpPool *p = new ExpPool();  // intentionally leaked: no destructor races at exit
        return *p;
    }
    void start(ExpJob *j, int helpers) {
        gate.lock();
        std::unique_lock<std::mutex> lk(m);
        for (; spawned < helpers; ++spawned) {
            const int id = spawned;
            std::thread([this, id] { loop(id); }).detach();
        }
        j->busy.store(helpers, std::memory_order_relaxed);
        job = j;
        wanted = helpers;
        ++generation;
        lk.unlock();
        cv.notify_all();
    }
    void finish(ExpJob *j) {
        while (j->busy.load(std::memory_order_acquire) > 0) std::this_thread::yield();
        {
            std::lock_guard<std::mutex> lk(m);
            job = nullptr;
        }
        gate.unlock();
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            ExpJob *j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return generation != seen; });
                seen = generation;
                j = id < wanted ? job : nullptr;
            }
            if (j) {
                j->work();
                j->busy.fetch_sub(1, std::memory_order_release);
            }
        }
    }
};
}  // namespace

namespace rlppo {
void set_exp_fast_transform(int on) { g_exp_fast = on != 0; }
}  // namespace rlppo

namespace {
// The stream phase: 2 n tempered MT19937 words, drawn exactly as c10::mt19937 would hand them out, into w; the serialised state
// advances in place.  `published` (may be null) is told how many ELEMENTS have both their words in w.
int exp_stream(TorchState *ts, int64_t n, uint32_t *w, std::atomic<int64_t> *published) {
    if (ts->left < 1 || ts->left > N || ts->next > (uint64_t)N) return RLPPO_ERR_ARG;
    static const bool have512 = __builtin_cpu_supports("avx512f") != 0;
    uint32_t mt[N + 16];
    for (int i = 0; i < N; i++) mt[i] = (uint32_t)ts->mt[i];
    int left = ts->left;
    uint32_t next = (uint32_t)ts->next;
    int64_t need = 2 * n, got = 0;
    while (got < need) {
        // words still unread in the current block: state[next .. 623] are readable while left - 1 > 0 reads remain
        int avail = left - 1;
        if (avail == 0) {  // `--left == 0` -> next_state(): left = 624, next = 0, and this call reads state[0]
            if (have512) regen512(mt);
            else regen(mt);
            left = N + 1;  // bookkeeping: after the read below left == 624, as in c10::mt19937
            next = 0;
            avail = N;
        }
        int take = (int64_t)avail < need - got ? avail : (int)(need - got);
        int k = have512 ? temper512(mt + next, w + got, take) : 0;
        for (; k + 8 <= take; k += 8) temper8(mt + next + k, w + got + k);
        for (; k < take; k++) {
            uint32_t y = mt[next + k];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            w[got + k] = y;
        }
        got += take;
        next += (uint32_t)take;
        left -= take;
        if (published) published->store(got / 2, std::memory_order_release);
    }
    for (int i = 0; i < N; i++) ts->mt[i] = mt[i];
    ts->left = left;
    ts->next = next;
    return 0;
}
}  // namespace

extern "C" int rlppo_torch_cpu_exponential(void *state, int64_t state_bytes, int64_t n, double lambda, float *out, int32_t threads) {
    if (!state || state_bytes < (int64_t)sizeof(TorchState) || n < 0 || (n > 0 && !out) || !(lambda > 0.0)) return RLPPO_ERR_ARG;
    if (n == 0) return 0;
    TorchState *ts = reinterpret_cast<TorchState *>(state);
    if (ts->left < 1 || ts->left > N || ts->next > (uint64_t)N) return RLPPO_ERR_ARG;

    // The stream phase (serial: MT19937 regeneration + tempering, ~0.9 ns per word) publishes the tempered words chunk by
    // chunk; the transform phase (double-precision log1p, ~7 ns per element) runs on persistent helper threads that claim
    // slices of 2048 elements as soon as their words are published -- the two phases overlap instead of adding up.
    static thread_local std::vector<uint32_t> words;
    words.resize((size_t)(2 * n) + 8);
    uint32_t *w = words.data();
    const double scale = -1.0 / lambda;
    ExpJob job;
    job.w = w;
    job.out = out;
    job.n = n;
    job.scale = scale;
    int t = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    if ((int64_t)(t - 1) * ExpJob::SLICE > n) t = (int)(n / ExpJob::SLICE) + 1;
    // With the vectorised transform (0.3-2 ns per element) the calling thread alone finishes the 4096 x 90 draw of a rollout step in
    // 0.4 ms; helpers only add their wake-up and hand-shake latency to a draw of that size (0.6 ms with 2 threads, 1.1 ms with 8 on
    // a busy host).  Large draws still spread over the pool.
    if (g_exp_fast) {
        const int by_size = 1 + (int)(n >> 20);  // one helper per 2^20 elements: none for the draw of a rollout step
        t = t < by_size ? t : by_size;
    }
    ExpPool &pool = ExpPool::get();
    if (t > 1) pool.start(&job, t - 1);
    const int rc = exp_stream(ts, n, w, &job.published);
    if (rc) job.published.store(n, std::memory_order_release);  // (cannot happen: the state was checked above) never leave helpers waiting
    job.work();                  // the calling thread helps with what is left
    if (t > 1) pool.finish(&job);
    return rc;
}

// [r3] The same draw as two calls, for a caller that pipelines consecutive draws (engine.HostExponential): the stream phase of draw
// k + 1 needs nothing from draw k but the generator state its stream phase left behind, so it can run while draw k is still being
// transformed on another thread.  rlppo_torch_cpu_exponential_words fills words[2 n (+ 8 of slack)] and advances the state exactly
// as the one-call form does; rlppo_exponential_from_words turns them into the n float32 values -- the identical values.
extern "C" int rlppo_torch_cpu_exponential_words(void *state, int64_t state_bytes, int64_t n, uint32_t *words) {
    if (!state || state_bytes < (int64_t)sizeof(TorchState) || n < 0 || (n > 0 && !words)) return RLPPO_ERR_ARG;
    if (n == 0) return 0;
    return exp_stream(reinterpret_cast<TorchState *>(state), n, words, nullptr);
}
// [r3] One draw as ONE call that can be chained to its predecessor without the caller in between: `link_in` (may be null) is the
// link block another thread's call publishes -- { int32 ready; pad to 64 bytes; state bytes } -- and this call spins (no GIL, no
// syscall) until that call's STREAM phase has finished, takes the state it left behind, runs its own stream phase, publishes ITS
// link block (`link_out`, required) and only then transforms.  Two threads alternating over consecutive draws therefore overlap
// the transform of draw k with the stream phase (and the transform) of draw k + 1: one draw leaves the pair every half draw time.
// With link_in == null the start state is `state` (not modified: the state after this draw is link_out's).
extern "C" int rlppo_torch_cpu_exponential_chained(const void *state, int64_t state_bytes, int64_t n, double lambda, float *out,
                                                   uint32_t *words, void *link_in, void *link_out) {
    if (state_bytes < (int64_t)sizeof(TorchState) || n <= 0 || !out || !words || !link_out || !(lambda > 0.0) || (!state && !link_in))
        return RLPPO_ERR_ARG;
    auto ready_of = [](void *l) { return reinterpret_cast<std::atomic<int32_t> *>(l); };
    auto state_of = [](void *l) { return reinterpret_cast<char *>(l) + 64; };
    std::vector<char> local((size_t)state_bytes);
    if (link_in) {
        uint64_t spins = 0;
        while (ready_of(link_in)->load(std::memory_order_acquire) == 0) {
            _mm_pause();
            if (++spins > (1ull << 28)) {  // ~10 s: the predecessor died without publishing (it normally takes 0.1 ms)
                ready_of(link_out)->store(-1, std::memory_order_release);
                return RLPPO_ERR_ARG;
            }
        }
        if (ready_of(link_in)->load(std::memory_order_acquire) < 0) {  // the predecessor failed: pass the failure on
            ready_of(link_out)->store(-1, std::memory_order_release);
            return RLPPO_ERR_ARG;
        }
        memcpy(local.data(), state_of(link_in), (size_t)state_bytes);
    } else {
        memcpy(local.data(), state, (size_t)state_bytes);
    }
    const int rc = exp_stream(reinterpret_cast<TorchState *>(local.data()), n, words, nullptr);
    if (rc) {
        ready_of(link_out)->store(-1, std::memory_order_release);
        return rc;
    }
    memcpy(state_of(link_out), local.data(), (size_t)state_bytes);
    ready_of(link_out)->store(1, std::memory_order_release);
    exp_transform(words, out, 0, n, -1.0 / lambda);
    return 0;
}
// [r4] A RUN of chained draws on the calling thread, for a caller that knows many draws ahead what it will need (the rollout of the
// next iteration: 128 draws of [4096, 90], which the two helper threads produce while PPOLearner.learn keeps the GPU busy and the
// collect critical path then only uploads).  Of a burst of `count` consecutive draws of n values each, this call performs draws
// first, first + step, first + 2 step, ... -- `step` threads, one call each with first = 0 .. step - 1, share the burst exactly as
// two alternating rlppo_torch_cpu_exponential_chained callers would, but with no hop through the caller between draws.
// Draw i writes out + i * out_stride (floats) and publishes links + i * link_stride (bytes): { int32 ready (stream phase done, state
// valid); int32 done (values complete; -1: failed or cancelled); padding to 64 bytes; state }.  Draw 0 starts from `state`, or from
// `link_in0` (a link block an earlier chained call publishes) when that is given.  `cancel`: a caller-owned word checked before
// every draw; once non-zero the remaining draws of this call are marked failed and the call returns.
extern "C" int rlppo_torch_cpu_exponential_burst(const void *state, void *link_in0, int64_t state_bytes, int64_t n, double lambda,
                                                 float *out, int64_t out_stride, void *links, int64_t link_stride, int32_t first,
                                                 int32_t step, int32_t count, const int32_t *cancel) {
    if (state_bytes < (int64_t)sizeof(TorchState) || n <= 0 || !out || !links || !(lambda > 0.0) || (!state && !link_in0) || first < 0 ||
        step < 1 || count < 1 || out_stride < n || link_stride < 64 + state_bytes || !cancel)
        return RLPPO_ERR_ARG;
    static thread_local std::vector<uint32_t> words;
    words.resize((size_t)(2 * n) + 8);
    auto link = [&](int i) { return reinterpret_cast<char *>(links) + (int64_t)i * link_stride; };
    auto done_of = [&](int i) { return reinterpret_cast<std::atomic<int32_t> *>(link(i) + 4); };
    auto ready_of = [&](int i) { return reinterpret_cast<std::atomic<int32_t> *>(link(i)); };
    int rc = 0;
    for (int i = first; i < count; i += step) {
        if (rc == 0 && __atomic_load_n(cancel, __ATOMIC_ACQUIRE) != 0) rc = RLPPO_ERR_ARG;
        if (rc == 0) {
            rc = rlppo_torch_cpu_exponential_chained(i == 0 && !link_in0 ? state : nullptr, state_bytes, n, lambda, out + (int64_t)i * out_stride,
                                                     words.data(), i == 0 ? link_in0 : link(i - 1), link(i));
        } else {
            ready_of(i)->store(-1, std::memory_order_release);  // the successor (another thread's run) must not wait for us
        }
        done_of(i)->store(rc == 0 ? 1 : -1, std::memory_order_release);
    }
    return rc;
}
extern "C" int rlppo_exponential_from_words(const uint32_t *words, int64_t n, double lambda, float *out) {
    if (n < 0 || (n > 0 && (!words || !out)) || !(lambda > 0.0)) return RLPPO_ERR_ARG;
    if (n > 0) exp_transform(words, out, 0, n, -1.0 / lambda);
    return 0;
}

// [r5] Host side of rlppo_act_opts.done_words: spins (no system call, no GIL: the Python host calls it through ctypes) until the
// `count` words all hold `value`, or `timeout_us` has passed (returns 1: the caller synchronises the stream instead).  The words live
// in pinned host memory the kernel stores into with release semantics at system scope; the acquire loads here order the reads of
// the results behind them.
// [r5] The windows rlppo_host_window_alloc has handed out, with the HDP flush register of their device.  A host write through the PCIe
// aperture passes through the GPU's host data path on its way to memory; a store of 1 to HDP_MEM_COHERENCY_FLUSH_CNTL -- which HIP maps
// into the process (hipDeviceProp_t::hdpMemFlushCntl) -- waits until everything in it has landed (rlppo_host_window_flush).
namespace {
struct HostWindow {
    uintptr_t base, end;
    volatile uint32_t *hdp_flush;
};
constexpr int MAX_WINDOWS = 256;
HostWindow g_windows[MAX_WINDOWS];
std::atomic<int> g_n_windows{0};
std::mutex g_windows_mutex;

inline void hdp_flush_for(const void *p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const int n = g_n_windows.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (a >= g_windows[i].base && a < g_windows[i].end) {
            if (g_windows[i].hdp_flush) {
                *g_windows[i].hdp_flush = 1u;
                _mm_sfence();
            }
            return;
        }
}
}  // namespace
void host_window_register(void *base, size_t bytes, unsigned *hdp_flush) {
    std::lock_guard<std::mutex> lock(g_windows_mutex);
    const int n = g_n_windows.load(std::memory_order_relaxed);
    for (int i = 0; i < n; ++i)
        if (g_windows[i].base == 0) {  // a freed slot
            g_windows[i].end = reinterpret_cast<uintptr_t>(base) + bytes;
            g_windows[i].hdp_flush = hdp_flush;
            g_windows[i].base = reinterpret_cast<uintptr_t>(base);
            return;
        }
    if (n < MAX_WINDOWS) {
        g_windows[n] = HostWindow{reinterpret_cast<uintptr_t>(base), reinterpret_cast<uintptr_t>(base) + bytes, hdp_flush};
        g_n_windows.store(n + 1, std::memory_order_release);
    }
}
void host_window_unregister(void *base) {
    std::lock_guard<std::mutex> lock(g_windows_mutex);
    const int n = g_n_windows.load(std::memory_order_relaxed);
    for (int i = 0; i < n; ++i)
        if (g_windows[i].base == reinterpret_cast<uintptr_t>(base)) g_windows[i].base = g_windows[i].end = 0;
}

// [r5] Host writes into DEVICE memory (rlppo_host_window_alloc: fine-grained VRAM behind the PCIe aperture, write-combined on the host
// side): a posted write costs the host 1.5 us per 3 KB / 2.5 us per 34 KB, where a GPU-initiated read of the same bytes from pinned
// host memory cost the rollout kernel 11-20 us (tools/get_action_profile.py).  The copy is fenced (write-combining buffers drain, and
// nothing after it -- the doorbell of a launch, the flag below -- can overtake it); the optional flag word follows, fenced again.
extern "C" int rlppo_host_push(void *dst, const void *src, size_t bytes, uint32_t *flag, uint32_t value) {
    if (bytes && (!dst || !src)) return RLPPO_ERR_ARG;
    if (bytes) {
        memcpy(dst, src, bytes);
        _mm_sfence();
    }
    if (flag) {
        __atomic_store_n(flag, value, __ATOMIC_RELAXED);
        _mm_sfence();
    }
    return 0;
}

// the host half of a small rollout call before its launch, in one call: control words {sequence, live rows} (ctl may be NULL) and
// the observations, one fence behind both
extern "C" int rlppo_host_stage_call(uint32_t *ctl, uint32_t sequence, uint32_t live_rows, void *obs_dst, const void *obs_src, size_t obs_bytes) {
    if (obs_bytes && (!obs_dst || !obs_src)) return RLPPO_ERR_ARG;
    if (ctl) {
        ctl[0] = sequence;
        ctl[1] = live_rows;
    }
    if (obs_bytes) memcpy(obs_dst, obs_src, obs_bytes);
    _mm_sfence();
    return 0;
}

// ... the same with the observations written as ROWS of a padded image: row r of `rows` (row_bytes each, src_stride apart) goes to
// dst + r * dst_stride -- a window that was zeroed once and only ever receives the first row_bytes of its rows IS the zero-padded
// input of the first layer (the layer chain of a small call then needs no pad launch)
extern "C" int rlppo_host_stage_rows(uint32_t *ctl, uint32_t sequence, uint32_t live_rows, void *dst, size_t dst_stride, const void *src,
                                     size_t src_stride, size_t row_bytes, int64_t rows) {
    if (rows < 0 || (rows > 0 && (!dst || !src || row_bytes > dst_stride || row_bytes > src_stride))) return RLPPO_ERR_ARG;
    if (ctl) {
        ctl[0] = sequence;
        ctl[1] = live_rows;
    }
    for (int64_t r = 0; r < rows; ++r)
        memcpy(static_cast<char *>(dst) + r * dst_stride, static_cast<const char *>(src) + r * src_stride, row_bytes);
    _mm_sfence();
    return 0;
}

// HDP flush of the device `window` lives on (see the windows' table above): every host write that has left the CPU is in device
// memory when this returns.  ~2.5 us (an uncached register write): ActGraph issues it BEHIND the launch, where it costs the call
// nothing -- the launch's own way to the first wave (doorbell, the packet fetched over PCIe, dispatch: >= 3 us) already orders the
// staged bytes before the kernel's reads, the flush closes the window from the other side.
extern "C" int rlppo_host_window_flush(const void *window) {
    if (!window) return RLPPO_ERR_ARG;
    hdp_flush_for(window);
    return 0;
}

extern "C" int rlppo_host_wait_words(const uint32_t *words, int64_t count, uint32_t value, int64_t timeout_us) {
    if (count < 0 || (count > 0 && !words)) return RLPPO_ERR_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    int64_t first = 0;
    for (unsigned spins = 0;; ++spins) {
        while (first < count) {
            const uint32_t w = __atomic_load_n(words + first, __ATOMIC_ACQUIRE);
            if (w == (value | 0x80000000u) && w != value) return 2;  // the kernel gave up on its late noise (rlppo_act_opts.noise_ctl)
            if (w != value) break;
            ++first;
        }
        if (first >= count) return 0;
        __builtin_ia32_pause();
        if ((spins & 1023) == 1023 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_us)
            return 1;
    }
}
