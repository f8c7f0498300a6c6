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
    if (!st || n < 0 || (n > 0 && !out) || n > 0xffffffffLL) return RLPPO_ERR_ARG;
    for (int64_t i = 0; i < n; i++) out[i] = i;
    for (int64_t i = n - 1; i >= 1; i--) {
        // random_interval(max = i): smallest all-ones mask >= i, redraw until <= i
        uint64_t mask = (uint64_t)i;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        uint64_t j;
        do {
            j = next32(st) & mask;
        } while (j > (uint64_t)i);
        const int64_t t = out[j];
        out[j] = out[i];
        out[i] = t;
    }
    return 0;
}
}
