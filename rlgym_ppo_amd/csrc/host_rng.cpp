// host_rng.cpp -- HOST code: numpy's legacy RandomState stream (MT19937, init_genrand seeding, 32-bit masked
// rejection sampling, reverse Fisher-Yates), i.e. the index stream of ExperienceBuffer.get_all_batches_shuffled
// (reference: rlgym_ppo/ppo/experience_buffer.py:52,97-98 -> numpy.random.RandomState(seed).permutation(n)).
// numpy is a third-party dependency of the reference (requirements.txt:7); the algorithm restated here is the
// published one of numpy/random/src/mt19937/mt19937.c and legacy-distributions.c (random_interval) and
// mtrand.pyx (_shuffle_raw), pinned by tests against numpy itself and by fixture tests/golden/g6_shuffle.npz.
// It exists so that the permutation of a 512k-sample buffer takes ~3 ms instead of numpy's ~12 ms and can run
// outside the GIL while the GPU works on the previous epoch.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/rlppo.h"

namespace {
constexpr int N = 624, M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

inline void regen(uint32_t *mt) {
    int kk;
    uint32_t y;
    for (kk = 0; kk < N - M; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; kk < N - 1; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
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
}
