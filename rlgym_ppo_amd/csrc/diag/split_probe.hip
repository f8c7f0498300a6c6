// split_probe.hip -- EXPERIMENT (librlppo_diag.so, not the product): the hidden-layer forward C = relu(A . W^T + b) with fp32 data in
// memory and the products on the bf16 MFMA pipe, every fp32 operand split into three bf16 pieces (x = xh + xm + xl EXACTLY: three
// 8-bit slices of the 24-bit significand, taken by truncation), six of the nine piece products kept:
//     x . w  ~=  xh.wh + xh.wm + xm.wh + xm.wm + xh.wl + xl.wh        (dropped: xm.wl, xl.wm, xl.wl <= 3 . 2^-24 |x||w|)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16, the small terms of a 32-wide K block first.  MI355X's fp32-input MFMA runs at
// 1/16 of the bf16 rate, so six bf16 MFMAs per product block are still 2.6 x its flops.
// Operands: A [M][K] fp32 as the update stores its activations -- split ON THE FLY, per fragment, in registers (4 VALU + 1.5 packing
// instructions per value); W pre-split at pack time into three bf16 planes in stage-major order [K / 32][3][N = 256][32] (what a
// round-5 rlppo_net_pack would emit).  One 256 x 256 output tile per workgroup (8 waves as 4 x 2, 128 accumulator registers per
// lane, one workgroup per CU), two stages of 80 KiB: A 256 x 32 fp32 (32 KiB, the 128-byte-row image and swizzle of the bf16 kernels)
// + three 16 KiB weight planes.  K == N == 256 only; M a multiple of 256.
#include "../gemm_detail.hpp"
#include "diag.hpp"

namespace rlppo {
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// three bf16x8 fragments (8 consecutive k of one row) from 8 fp32 values.  RNE = false: pieces by truncation (x = h + m + l exactly;
// |m| <= 2^-8 |x|, |l| <= 2^-16 |x|), packed with v_perm.  RNE = true: pieces rounded to nearest even by v_cvt_pk_bf16_f32 (signed
// residuals: |m| <= 2^-9 |x|, |l| <= 2^-18 |x|, so the dropped products are 8 x smaller; x = h + m + l up to 2^-26 |x|).
template <bool RNE>
__device__ __forceinline__ void split8(const f32x4 &lo, const f32x4 &hi, bf16x8 &ph, bf16x8 &pm, bf16x8 &pl) {
    u32x4 h, m, l;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = p < 2 ? lo[2 * p] : hi[2 * p - 4], x1 = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
        if (RNE) {
            const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
            const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);   // exact
            const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
            const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);   // exact
            h[p] = hp;
            m[p] = mp;
            l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
        } else {
            const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
            const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);  // exact
            const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
            const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);  // exact, <= 8 bits left
            h[p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);   // [x1.hi16 | x0.hi16]: element 2p in the low half
            m[p] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
            l[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
        }
    }
    ph = __builtin_bit_cast(bf16x8, h);
    pm = __builtin_bit_cast(bf16x8, m);
    pl = __builtin_bit_cast(bf16x8, l);
}

constexpr int TM = 256, TN = 256, BK = 32;            // tile, K step
constexpr int A_STAGE = TM * BK * 4;                  // 32 KiB
constexpr int W_PLANE = TN * BK * 2;                  // 16 KiB
constexpr int STAGE = A_STAGE + 3 * W_PLANE;          // 80 KiB

template <int TERMS, bool RNE, bool SEP = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_split_kernel(const float *__restrict__ A, unsigned lda_b, const unsigned short *__restrict__ Ws,
                                                               const float *__restrict__ bias, float *__restrict__ C, unsigned ldc_b,
                                                               int64_t M, int K, int store) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [2][A 256 x 128 B swizzled | 3 x W plane 256 x 64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wr = wave_u >> 1, wc = wave_u & 1;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b, (unsigned)(TM - 1) * lda_b + (unsigned)K * 4);
    const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(Ws, (unsigned)nk * (3u * W_PLANE));
    // A pieces: 8 rows x 128 B per instruction (lane -> row lane / 8, chunk lane % 8, swizzled on the source side), 4 per wave and stage
    const int row_p = wave * 8 + lane / 8, pch = lane % 8, lch = pch ^ (row_p & 7);
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    auto issue = [&](int buf, int kt) {
        char *Ad = lds + buf * STAGE + wave_u * (8 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, reinterpret_cast<float *>(Ad + i * (64 * 128)), 16, a_off, (unsigned)kt * 128u + i * 64u * lda_b, 0, 0);
        char *Wd = lds + buf * STAGE + A_STAGE + wave_u * (6 * 1024);  // the stage's 48 KiB of weights are contiguous in memory: 6 KiB per wave
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, reinterpret_cast<float *>(Wd + i * 1024), 16, (unsigned)lane * 16u,
                                                     (unsigned)kt * (3u * W_PLANE) + (unsigned)(wave_u * 6 + i) * 1024u, 0, 0);
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[0][j] = *reinterpret_cast<const f32x4 *>(&bias[wc * 128 + j * 16 + q * 4]);
        acc[1][j] = acc[2][j] = acc[3][j] = acc[0][j];
    }
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(cur ^ 1, kt + 1);
        const char *Ac = lds + cur * STAGE;
        const char *Wc = Ac + A_STAGE;
        bf16x8 ah[4], am[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wr * 64 + i * 16 + r16;
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(Ac + row * 128 + (((2 * q) ^ (row & 7)) * 16));
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(Ac + row * 128 + (((2 * q + 1) ^ (row & 7)) * 16));
            split8<RNE>(lo, hi, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            bf16x8 wh[4], wm[4], wl[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int col = wc * 128 + (jh * 4 + jj) * 16 + r16;
                wh[jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + col * 64 + q * 16));
                wm[jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + W_PLANE + col * 64 + q * 16));
                wl[jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + 2 * W_PLANE + col * 64 + q * 16));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (SEP) {
                        // the five small products of this K block are summed among THEMSELVES first (a chain that starts from 0: its
                        // roundings are 2^-8 of the main product's) and enter the accumulator with one addition; the accumulator
                        // itself sees one MFMA and one add per block instead of six MFMAs
                        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], al[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[jj], ah[i], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], am[i], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], am[i], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], ah[i], t, 0, 0, 0);
                        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], ah[i], acc[i][jh * 4 + jj], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) c[e] += t[e];
                        acc[i][jh * 4 + jj] = c;
                        continue;
                    }
                    f32x4 c = acc[i][jh * 4 + jj];  // small terms first
                    if (TERMS >= 8) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], al[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[jj], am[i], c, 0, 0, 0);
                    }
                    if (TERMS >= 6) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], al[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[jj], ah[i], c, 0, 0, 0);
                    }
                    if (TERMS >= 4) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], am[i], c, 0, 0, 0);
                    if (TERMS >= 3) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], am[i], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], ah[i], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], ah[i], c, 0, 0, 0);
                    acc[i][jh * 4 + jj] = c;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!store) return;
    // relu, fp32 out: 16 bytes per lane (4 consecutive columns of one row), as gemm_nt_dma_kernel stores them
    const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)(wc * 128) * 4, (unsigned)(TM - 1) * ldc_b + 128 * 4);
    const unsigned c_off = (unsigned)(wr * 64 + r16) * ldc_b + q * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rs, c_off, 16 * i * ldc_b + j * 64, 0);
        }
}
}  // namespace

int launch_gemm_nt_split(hipStream_t st, const float *A, int64_t lda, const unsigned short *Ws, const float *bias, float *C, int64_t ldc,
                         int64_t M, int N, int K, int terms, int store) {
    const bool sep = terms >= 200;  // terms + 200: 6 products, nearest pieces, the five small ones summed apart from the accumulator
    const bool rne = terms >= 100;  // terms + 100: the activations' pieces rounded to nearest (the weight planes are the caller's)
    terms %= 100;
    RLPPO_CHECK_ARG(N == 256 && K % 32 == 0 && K >= 32 && M > 0 && M % 256 == 0 && A && Ws && bias && C && lda >= K && ldc >= N &&
                        (terms == 1 || terms == 3 || terms == 4 || terms == 6 || terms == 8),
                    "gemm_nt_split: N=%d K=%d M=%ld terms=%d", N, K, (long)M, terms);
    static PerDeviceOnce attr[10];
    constexpr int LDS_BYTES = 2 * STAGE;
    const dim3 grid((unsigned)(M / 256));
#define SPLIT3(T, R, S, SLOT)                                                                                                 \
    do {                                                                                                                      \
        if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_split_kernel<T, R, S>, LDS_BYTES, attr[SLOT])) return rc_;      \
        hipLaunchKernelGGL((gemm_nt_split_kernel<T, R, S>), grid, dim3(512), LDS_BYTES, st, A, (unsigned)(lda * 4), Ws, bias, C, \
                           (unsigned)(ldc * 4), M, K, store);                                                                 \
    } while (0)
#define SPLIT(T, R, SLOT) SPLIT3(T, R, false, SLOT)
    if (sep) {
        SPLIT3(6, true, true, 9);
        RLPPO_LAUNCH_CHECK();
        return 0;
    }
    switch (terms + (rne ? 100 : 0)) {
        case 1: SPLIT(1, false, 0); break;
        case 3: SPLIT(3, false, 1); break;
        case 4: SPLIT(4, false, 2); break;
        case 6: SPLIT(6, false, 3); break;
        case 8: SPLIT(8, false, 4); break;
        case 106: SPLIT(6, true, 5); break;
        case 108: SPLIT(8, true, 6); break;
        case 104: SPLIT(4, true, 7); break;
        default: SPLIT(3, true, 8); break;
    }
#undef SPLIT
#undef SPLIT3
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo
