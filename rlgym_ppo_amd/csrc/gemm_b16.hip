// gemm_b16.hip -- the GEMM kernels of the bf16 UPDATE precision (rlppo_set_update_precision(1)): both
// operands bf16 IN MEMORY, v_mfma_f32_16x16x32_bf16, fp32 accumulation.
//   gemm_nt_b16 / gemm_nt_b16w : hidden-layer forward (relu, one rounding, bf16 + ReLU bitmask out), output layer (fp32 out), and the
//                                backward product dX = round_bf16(dY . W) masked by the forward's bitmask
//   gemm_tn_b16 / gemm_tn_b16w : dW = dY^T . X, contraction over rows through the transposing LDS read (ds_read_b64_tr_b16)
// The `w` forms work on 256 x 256 output tiles (one workgroup of 8 waves per CU, 128 KiB of dynamic LDS).  Staging, swizzles, bitmask
// layout, partial-tile layout and the reduction are those of the fp32 kernels (gemm.hip, gemm_detail.hpp).
#include "gemm_detail.hpp"
// -DB16_ABL=bits builds TIMING-ONLY variants of the persistent kernel (results are garbage) for tools/b16_ablation.py: 1 the K steps'
// stage requests go outside their descriptors (issued and counted, no traffic), 16 the tile's stores dropped, 32 no epilogue work.
// [r6] pricing of a weight-stationary form: 2 only the WEIGHT half of every stage request dropped (what a workgroup that kept its
// column slab of W resident would not stage), 4 only the activation half, 8 the weight fragments are not read from LDS either (the
// registers loaded at the tile's first K step are multiplied again and again) -- 2 + 8 = the ceiling of ANY weight-stationary kernel
// with this tile: weights neither staged nor read.
#ifndef B16_ABL
#define B16_ABL 0
#endif

namespace rlppo {

// ------------------------------------------------------------------------------------------------ gemm_nt, bf16 in memory
// Forward product of the bf16 UPDATE precision (rlppo_set_update_precision(1); BASELINE configs[4] "bf16 fwd / fp32 master
// weights"): both operands are bf16 IN MEMORY -- the activations as the previous layer's epilogue (or the minibatch gather)
// left them, the weights as rlppo_net_pack_bf16 rounded the fp32 master copy after the optimiser step -- so a tile row of
// 64 k-values is the same 128 bytes as 32 fp32 values: staging, LDS image and swizzle are those of gemm_nt_dma_kernel<.., 32>,
// the K loop runs half as many tiles, a ds_read_b128 fragment feeds ONE v_mfma_f32_16x16x32_bf16 (no conversion in the loop),
// accumulation, bias and activation stay fp32.
// HIDDEN layers: h = relu(acc) is rounded to bf16 once (v_cvt_pk_bf16_f32, round-to-nearest-even) and written three ways:
// as bf16 (the next layer's A operand, 2 B/element), as the same value in fp32 (the X operand of the fp32 weight-gradient
// product: autograd of a forward with bf16-rounded operands multiplies dY with the ROUNDED input), and as the ReLU bitmask
// of the dX product.  The output layer stores plain fp32 (bias or bias + tanh) for the loss kernel.
// B16_DX is the backward product of the same precision, dX = (dY . W) rounded to bf16 and masked by relu'(h): in mixed-precision
// training the hidden activations are bf16 tensors, so the gradient with respect to each of them is a bf16 tensor too
//  -- A = dY[M][pout] bf16, B = W^T[pin][pout] bf16, no bias, the mask is the ReLU bitmask the
// forward left for this tile geometry (one 8-byte word per lane, read before the K loop), the output is bf16 only.
enum { B16_OUT = 0, B16_HIDDEN = 1, B16_DX = 2 };
template <int NB, int EPI, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_b16_kernel(const unsigned short *__restrict__ A, unsigned lda_b,
                                                             const unsigned short *__restrict__ B, unsigned ldb_b,
                                                             const float *__restrict__ bias, float *__restrict__ C,
                                                             unsigned ldc_b, unsigned short *__restrict__ Cb, unsigned ldcb_b,
                                                             int64_t M, int K, unsigned long long *__restrict__ bits) {
    constexpr int BN = NB * 16;
    constexpr int BKT = 32;          // tile row = 128 bytes = 64 bf16 (the fp32 kernel's 32 floats)
    constexpr int CPR = BKT / 4, RPW = 64 / CPR, RPP = 4 * RPW;
    constexpr int A_IT = SBM / RPP, B_IT = BN / RPP;
    static_assert(BN % RPP == 0, "column tile must be a whole number of staging passes");
    constexpr bool HIDDEN = MODE == B16_HIDDEN, DX = MODE == B16_DX;
    static_assert(!HIDDEN || EPI == EPI_BIAS_RELU, "hidden layers are bias + ReLU");
    static_assert(!DX || (EPI == EPI_MASK && NB == 8), "the dX form masks whole 128 x 128 tiles");
    __shared__ __attribute__((aligned(16))) float lds[2 * SBM * BKT + 2 * BN * BKT];
    float *As = lds;
    float *Bs = lds + 2 * SBM * BKT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    int row_tile, col_tile;
    xcd_tile(row_tile, col_tile);
    const int64_t m0 = (int64_t)row_tile * SBM;
    const int n0 = col_tile * BN;
    const int rows_here = (int)((M - m0) < SBM ? (M - m0) : SBM);

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b,
                                                  (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 2);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(reinterpret_cast<const char *>(B) + (int64_t)n0 * ldb_b,
                                                  (unsigned)(BN - 1) * ldb_b + (unsigned)K * 2);
    const int row_p = wave * RPW + lane / CPR, pch = lane % CPR;
    const int lch = BKT == 32 ? (pch ^ (row_p & 7)) : (pch ^ ((0 - (row_p >> 2)) & 3));  // source side of dswz<BKT>
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    const unsigned b_off = (unsigned)row_p * ldb_b + lch * 16;
    const unsigned a_step = (unsigned)RPP * lda_b, b_step = (unsigned)RPP * ldb_b;

    f32x4 acc[2][NB];
    unsigned long long mask_word = 0;
    if (DX) {
        mask_word = bits[((size_t)row_tile * gridDim.y + col_tile) * 256 + tid];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        const __amdgpu_buffer_rsrc_t bias_rs = make_rsrc(bias + n0, BN * 4);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            acc[0][j] = ldb(bias_rs, (unsigned)(q * 16), j * 64);
            acc[1][j] = acc[0][j];
        }
    }
    auto issue_tile = [&](int buf, unsigned kb) {
        float *Ad = As + buf * SBM * BKT + wave_u * RPW * BKT;
        float *Bd = Bs + buf * BN * BKT + wave_u * RPW * BKT;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, Ad + i * RPP * BKT, 16, a_off, kb + i * a_step, 0, 0);
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, Bd + i * RPP * BKT, 16, b_off, kb + i * b_step, 0, 0);
    };

    const int nk = K / (2 * BKT);
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if ((kt + 1) < nk) issue_tile(cur ^ 1, (unsigned)(kt + 1) * (4u * BKT));
        const float *Ac = As + cur * SBM * BKT + (wave * 32) * BKT;
        const float *Bc = Bs + cur * BN * BKT;
#pragma unroll
        for (int kc = 0; kc < BKT / 16; ++kc) {  // 32 k-values per MFMA: chunk kc * 4 + q holds k = 32 kc + 8 q .. + 7 of the row
            bf16x8 fa[2], fb[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Ac[dswz<BKT>(i * 16 + r16, kc * 4 + q)]));
#pragma unroll
            for (int j = 0; j < NB; ++j)
                fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bc[dswz<BKT>(j * 16 + r16, kc * 4 + q)]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (MODE == B16_OUT) {
        nt_epilogue<NB, EPI>(acc, nullptr, 0, C, ldc_b, m0, n0, rows_here, wave, r16, q);
        return;
    }
    u32x2 pk[2][NB];
    if (DX) {  // one rounding to bf16 (the gradient of a bf16 activation), then relu'(h) from the forward's bitmask
        const unsigned lo = (unsigned)mask_word, hi = (unsigned)(mask_word >> 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                unsigned h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int b = (i * NB + j) * 4 + e;
                    const unsigned m = 0u - (((b < 32 ? lo : hi) >> (b & 31)) & 1u);  // 0 or ~0
                    const float x = acc[i][j][e];
                    h[e] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x) & m;
                }
                pk[i][j] = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
            }
        __builtin_amdgcn_sched_barrier(0);
        const __amdgpu_buffer_rsrc_t cb_rs = make_rsrc(reinterpret_cast<char *>(Cb) + m0 * ldcb_b + (int64_t)n0 * 2,
                                                       (unsigned)(rows_here - 1) * ldcb_b + BN * 2);
        const unsigned cb_off = (unsigned)(wave * 32 + r16) * ldcb_b + q * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) __builtin_amdgcn_raw_buffer_store_b64(pk[i][j], cb_rs, cb_off, 16 * i * ldcb_b + j * 32, 0);
        return;
    }
    // relu, bitmask, one rounding to bf16; the stores come last, from registers nothing writes again (section 5 hazard rule)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = relu1(acc[i][j][e]);
    const unsigned long long word = relu_bits<NB>(acc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            unsigned short h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = acc[i][j][e];
                h[e] = __builtin_bit_cast(unsigned short, (__bf16)x);
                acc[i][j][e] = __uint_as_float((unsigned)h[e] << 16);  // the rounded value, exactly, as fp32
            }
            pk[i][j] = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        }
    if (bits) bits[((size_t)row_tile * gridDim.y + col_tile) * 256 + tid] = word;
    __builtin_amdgcn_sched_barrier(0);
    const int row_l = wave * 32 + r16;
    if (Cb) {
        const __amdgpu_buffer_rsrc_t cb_rs = make_rsrc(reinterpret_cast<char *>(Cb) + m0 * ldcb_b + (int64_t)n0 * 2,
                                                       (unsigned)(rows_here - 1) * ldcb_b + BN * 2);
        const unsigned cb_off = (unsigned)row_l * ldcb_b + q * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) __builtin_amdgcn_raw_buffer_store_b64(pk[i][j], cb_rs, cb_off, 16 * i * ldcb_b + j * 32, 0);
    }
    if (C) {
        const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)n0 * 4,
                                                      (unsigned)(rows_here - 1) * ldc_b + BN * 4);
        const unsigned c_off = (unsigned)row_l * ldc_b + q * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) stb(c_rs, c_off, 16 * i * ldc_b + j * 64, acc[i][j]);
    }
}

// The same two products on a 256 x 256 output tile (8 waves as 4 x 2, each 64 rows x 128 columns).  The bf16 MFMA does 16x the
// flops per cycle of the fp32 one, so with 128 x 128 tiles the kernel above is bound by the rate at which tiles can be brought
// into LDS (the bytes in flight are capped by the LDS, ~2 us of latency each): 4.3 GB of tile traffic per 512 -> 512 launch of
// 524,288 rows at ~9.4 TB/s.  A 256 x 256 tile needs half the bytes per flop.  Two 64 KiB stages (dynamic LDS, one workgroup per
// CU); the ReLU bitmask keeps the 128 x 128-tile layout of relu_bits (a wave owns two of its 32-row lane slots), so every
// consumer of the bitmask is unchanged.
template <int MODE, int TM, int BKT, int NBUF>
__global__ __launch_bounds__(TM * 2, TM == 256 ? 1 : 2) void gemm_nt_b16w_kernel(const unsigned short *__restrict__ A, unsigned lda_b,
                                                              const unsigned short *__restrict__ B, unsigned ldb_b,
                                                              const float *__restrict__ bias, float *__restrict__ C, unsigned ldc_b,
                                                              unsigned short *__restrict__ Cb, unsigned ldcb_b, int64_t M, int K,
                                                              unsigned long long *__restrict__ bits, int row_tiles128) {
    // K tiles of 2 BKT values (tile rows of 4 BKT bytes) in a ring of NBUF stages: NBUF - 1 tiles are in flight while one is
    // multiplied.  Measured at 512 -> 512, 524,288 rows: two 64 KiB stages (BKT 32) ~400 us, four 32 KiB stages (BKT 16) ~440 us; the
    // 128 x 128 kernel above 458 us at two workgroups per CU and ~430 us with 32 KiB stages at four per CU.  In every form the MFMA pipe
    // is ~1/3 busy and the waves wait half of their cycles (SQ_WAIT_INST_ANY): after each barrier all 8 waves read their fragments
    // at once, and with one workgroup per CU nothing else fills that gap.
    // (TM = 128 -- 4 waves on a 128 x 256 tile with BKT = 16, two workgroups per CU so that one's epilogue runs under the other's K
    // loop -- was measured at 426 us against 399 us and is not instantiated.)
    constexpr int TN = 256, NW = TM / 32, STAGE = (TM + TN) * BKT;   // waves; floats per stage
    constexpr int CPR = BKT / 4, RPI = 64 / CPR, PASS = NW * RPI;    // 16-byte chunks per row, rows per wave instruction, rows per pass
    constexpr int DIST = NBUF - 1, PER_STAGE = (TM + TN) / PASS;     // DMA instructions per wave and stage
    constexpr bool DX = MODE == B16_DX;
    extern __shared__ __attribute__((aligned(16))) float wlds[];  // [NBUF][A: TM x BKT | B: TN x BKT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wr = wave_u >> 1, wc = wave_u & 1;
    int row_tile, col_tile;
    xcd_tile(row_tile, col_tile);
    const int64_t m0 = (int64_t)row_tile * TM;
    const int n0 = col_tile * TN;
    const int rows_here = (int)((M - m0) < TM ? (M - m0) : TM);

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b,
                                                  (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 2);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(reinterpret_cast<const char *>(B) + (int64_t)n0 * ldb_b,
                                                  (unsigned)(TN - 1) * ldb_b + (unsigned)K * 2);
    // DMA lane map: one wave instruction = RPI rows of 4 BKT bytes; a pass of the 8 waves = PASS rows
    const int row_p = wave * RPI + lane / CPR, pch = lane % CPR;
    const int lch = BKT == 32 ? (pch ^ (row_p & 7)) : (pch ^ ((0 - (row_p >> 2)) & 3));  // the source side of dswz<BKT>
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    const unsigned b_off = (unsigned)row_p * ldb_b + lch * 16;
    const unsigned a_step = (unsigned)PASS * lda_b, b_step = (unsigned)PASS * ldb_b;

    // bitmask words of this wave's two 32-row lane slots (128 x 128-tile layout of relu_bits)
    size_t widx[2];
    bool wlive[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rt128 = row_tile * (TM / 128) + (wr >> 1), ct128 = col_tile * 2 + wc, nct128 = gridDim.y * 2;
        widx[h] = ((size_t)rt128 * nct128 + ct128) * 256 + ((2 * wr + h) & 3) * 64 + lane;
        wlive[h] = rt128 < row_tiles128;
    }
    f32x4 acc[4][8];
    unsigned long long mask_word[2] = {0, 0};
    if (DX) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (wlive[h]) mask_word[h] = bits[widx[h]];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        const __amdgpu_buffer_rsrc_t bias_rs = make_rsrc(bias + n0 + wc * 128, 128 * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0][j] = ldb(bias_rs, (unsigned)(q * 16), j * 64);
            acc[1][j] = acc[2][j] = acc[3][j] = acc[0][j];
        }
    }
    auto issue_tile = [&](int buf, unsigned kb) {
        float *Ad = wlds + buf * STAGE + wave_u * RPI * BKT;
        float *Bd = wlds + buf * STAGE + TM * BKT + wave_u * RPI * BKT;
#pragma unroll
        for (int i = 0; i < TM / PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, Ad + i * PASS * BKT, 16, a_off, kb + i * a_step, 0, 0);
#pragma unroll
        for (int i = 0; i < TN / PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, Bd + i * PASS * BKT, 16, b_off, kb + i * b_step, 0, 0);
    };

    const int nk = K / (2 * BKT);
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < nk) issue_tile(d, (unsigned)d * (4u * BKT));
    int cur = 0, nxt = DIST % NBUF;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed when at most the later tiles' instructions are outstanding
        const int later = (nk - 1 - kt) < (DIST - 1) ? (nk - 1 - kt) : (DIST - 1);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_STAGE) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // everybody's pieces of tile kt are in LDS, and everybody is done reading tile kt - 1
        if (kt + DIST < nk) issue_tile(nxt, (unsigned)(kt + DIST) * (4u * BKT));  // into the buffer tile kt - 1 occupied
        const float *Ac = wlds + cur * STAGE + (wr * 64) * BKT;
        const float *Bc = wlds + cur * STAGE + TM * BKT + (wc * 128) * BKT;
#pragma unroll
        for (int kc = 0; kc < BKT / 16; ++kc) {
            bf16x8 fa[4], fb[8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Ac[dswz<BKT>(i * 16 + r16, kc * 4 + q)]));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bc[dswz<BKT>(j * 16 + r16, kc * 4 + q)]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }

    // epilogue, 32-row slot by slot (h): one rounding to bf16; forward: relu + bitmask first, dX: the forward's mask afterwards
    const int row_l = wr * 64 + r16;
    u32x2 pk[4][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned lo, hi;
        if (DX) {
            lo = (unsigned)mask_word[h];
            hi = (unsigned)(mask_word[h] >> 32);
        } else {
            lo = hi = 0;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = relu1(acc[2 * h + ii][j][e]);
                        acc[2 * h + ii][j][e] = x;
                        const int b = (ii * 8 + j) * 4 + e;
                        const unsigned v = x > 0.f ? 1u : 0u;
                        if (b < 32) lo |= v << b;
                        else hi |= v << (b - 32);
                    }
            if (bits && wlive[h]) bits[widx[h]] = ((unsigned long long)hi << 32) | lo;
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[2 * h + ii][j][e];
                    unsigned v = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
                    if (DX) {
                        const int b = (ii * 8 + j) * 4 + e;
                        v &= 0u - (((b < 32 ? lo : hi) >> (b & 31)) & 1u);
                    } else {
                        acc[2 * h + ii][j][e] = __uint_as_float(v << 16);  // the rounded value as fp32 (for the optional fp32 copy)
                    }
                    hv[e] = v;
                }
                pk[2 * h + ii][j] = u32x2{hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16)};
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        // The bf16 tile leaves through LDS: as it stands a wave-instruction would store 16 rows x 32 bytes (4 lanes x 8 bytes per
        // row), a shape the memory system writes at ~2.7 TB/s (the K = 64 launch of tools/b16_k_sweep.py: 570 MB in 210 us whatever the
        // tile shape).  Each wave parks its 64 x 128 sub-tile in its own 16 KiB of the (now idle) stage memory, 16-byte chunk c of
        // row r at chunk c ^ (r & 15), and reads it back row by row: 16 bytes per lane, 256 contiguous bytes per row, 4 rows per
        // wave-instruction.
        __syncthreads();  // every wave is done with the last stage
        char *mine = reinterpret_cast<char *>(wlds) + wave_u * (64 * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 16 * i + r16, ch = (2 * j + (q >> 1)) ^ (r & 15);
                *reinterpret_cast<u32x2 *>(mine + r * 256 + ch * 16 + 8 * (q & 1)) = pk[i][j];
            }
        // the wave reads only what it wrote itself: LDS operations of one wave complete in order, no barrier needed
        const int rr = lane >> 4, c16 = lane & 15;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int r = 4 * t + rr;
            v[t] = *reinterpret_cast<const u32x4 *>(mine + r * 256 + ((c16 ^ (r & 15)) * 16));
        }
        const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(reinterpret_cast<char *>(Cb) + m0 * ldcb_b + (int64_t)(n0 + wc * 128) * 2,
                                                      (unsigned)(rows_here - 1) * ldcb_b + 128 * 2);
        const unsigned o_off = (unsigned)(wr * 64 + rr) * ldcb_b + c16 * 16;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            __builtin_amdgcn_raw_buffer_store_b128(v[t], o_rs, o_off, 4 * t * ldcb_b, 0);
    }
    if (!DX && C) {
        const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)(n0 + wc * 128) * 4,
                                                      (unsigned)(rows_here - 1) * ldc_b + 128 * 4);
        const unsigned c_off = (unsigned)row_l * ldc_b + q * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) stb(c_rs, c_off, 16 * i * ldc_b + j * 64, acc[i][j]);
    }
}

// ------------------------------------------------------------------------------------------------ persistent form [r3]
// The 256 x 256-tile kernel above spends HALF of a 512 -> 512 launch outside its K loop (tools/b16_k_sweep.py: the intercept of
// time against K is 176 us of 351): with one workgroup per CU, a tile is [dispatch + bias / descriptors] -> [first stage's DMA
// latency, nothing to multiply] -> K loop -> [relu / bitmask / rounding: ~800 vector instructions per lane] -> [park in LDS, read
// back, store] -> workgroup exit, every bracket with the CU's MFMA pipe idle, 16 tiles per CU and launch (~11 us per tile).
// Here ONE workgroup per CU stays resident and walks its tiles (same column tile every time: its 256 rows of W stay in L2, bias
// and B descriptor are set up once); the first stage of tile t + 1 is requested BEFORE the epilogue of tile t starts, into the
// stage buffer the epilogue does not use -- the tile is parked in two halves through the other one (8 KiB per wave) -- so the
// DMA latency, the epilogue's vector work and its stores overlap, and the next K loop starts on data that is already in LDS.
// Counted waits: at the first K step of a tile the wave's stage-0 pieces are complete when all but the younger operations --
// the previous epilogue's 2 bitmask accesses and 16 tile stores -- are (vmcnt(18)); the first tile of a workgroup waits for
// everything.  XCD-aware walk: workgroup b owns column tile (b % (8 nc)) / 8 and row tiles 8 (b / (8 nc)) + b % 8 + k * (rows
// covered by the grid): the nc workgroups that share a row tile's A rows carry ids 8 apart, i.e. run on the same XCD.
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt_b16p_kernel(const unsigned short *__restrict__ A, unsigned lda_b,
                                                              const unsigned short *__restrict__ B, unsigned ldb_b,
                                                              const float *__restrict__ bias, unsigned short *__restrict__ Cb,
                                                              unsigned ldcb_b, int64_t M, int K, unsigned long long *__restrict__ bits,
                                                              int row_tiles, int col_tiles, int row_tiles128) {
    constexpr int TM = 256, TN = 256, BKT = 32, NW = 8, STAGE = (TM + TN) * BKT;  // floats per stage (64 KiB)
    constexpr int CPR = BKT / 4, RPI = 64 / CPR, PASS = NW * RPI;                    // 8 chunks per row, 8 rows per instruction, 64 per pass
    constexpr bool DX = MODE == B16_DX;
    extern __shared__ __attribute__((aligned(16))) float wlds[];  // [2][A: TM x BKT | B: TN x BKT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wr = wave_u >> 1, wc = wave_u & 1;
    // this workgroup's column tile and its walk over the row tiles
    const int b = blockIdx.x, group = 8 * col_tiles;
    const int col_tile = (b % group) >> 3;
    const int row_first = (b / group) * 8 + (b & 7), row_stride = (gridDim.x / group) * 8;
    const int n0 = col_tile * TN;

    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(reinterpret_cast<const char *>(B) + (int64_t)n0 * ldb_b,
                                                  (unsigned)(TN - 1) * ldb_b + (unsigned)K * 2);
    const int row_p = wave * RPI + lane / CPR, pch = lane % CPR;
    const int lch = pch ^ (row_p & 7);  // the source side of dswz<32>
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    const unsigned b_off = (unsigned)row_p * ldb_b + lch * 16;
    const unsigned a_step = (unsigned)PASS * lda_b, b_step = (unsigned)PASS * ldb_b;
    float *bias_l = wlds + 2 * STAGE;  // [256]: this column tile's bias (1 KiB behind the stages; registers are all taken)
    if (!DX) {
        if (tid < TN) bias_l[tid] = bias[n0 + tid];
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t bits_rs = make_rsrc(bits, (unsigned)((size_t)row_tiles128 * (col_tiles * 2) * 256 * 8));
    auto a_rsrc = [&](int row_tile) {
        const int64_t m0 = (int64_t)row_tile * TM;
        const int rows_here = (int)((M - m0) < TM ? (M - m0) : TM);
        return make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b, (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 2);
    };
    auto issue_tile = [&](const __amdgpu_buffer_rsrc_t &a_rs, int buf, unsigned kb) {
        float *Ad = wlds + buf * STAGE + wave_u * RPI * BKT;
        float *Bd = wlds + buf * STAGE + TM * BKT + wave_u * RPI * BKT;
        const unsigned kba = ((B16_ABL & 4) && kb != 0) ? 0x80000000u : kb, kbb = ((B16_ABL & 2) && kb != 0) ? 0x80000000u : kb;
#pragma unroll
        for (int i = 0; i < TM / PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, Ad + i * PASS * BKT, 16, a_off, kba + i * a_step, 0, 0);
#pragma unroll
        for (int i = 0; i < TN / PASS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, Bd + i * PASS * BKT, 16, b_off, kbb + i * b_step, 0, 0);
    };
    // byte offsets of this wave's two bitmask words of a row tile (128 x 128-tile layout of relu_bits); a slot past the last
    // 128-row tile gets an offset outside the descriptor: the access is dropped but still counted
    auto bits_off = [&](int row_tile, int h) -> unsigned {
        const int rt128 = row_tile * 2 + (wr >> 1), ct128 = col_tile * 2 + wc, nct128 = col_tiles * 2;
        return rt128 < row_tiles128 ? (unsigned)((((size_t)rt128 * nct128 + ct128) * 256 + ((2 * wr + h) & 3) * 64 + lane) * 8) : 0xFFFFFFF0u;
    };

    const int nk = K / (2 * BKT);
    if (row_first >= row_tiles) return;  // (a grid larger than the work: nothing was issued)
    __amdgpu_buffer_rsrc_t a_rs = a_rsrc(row_first);
    issue_tile(a_rs, 0, 0);
    bool first = true;
    u32x2 mask_next[2];
    if (DX) {
#pragma unroll
        for (int h = 0; h < 2; ++h) mask_next[h] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bits_rs, bits_off(row_first, h), 0, 0));
    }
    for (int row_tile = row_first; row_tile < row_tiles; row_tile += row_stride) {
        const int64_t m0 = (int64_t)row_tile * TM;
        const int rows_here = (int)((M - m0) < TM ? (M - m0) : TM);
        f32x4 acc[4][8];
        u32x2 mask_word[2];
        if (DX) {
#pragma unroll
            for (int h = 0; h < 2; ++h) mask_word[h] = mask_next[h];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[0][j] = *reinterpret_cast<const f32x4 *>(&bias_l[wc * 128 + j * 16 + q * 4]);
                acc[1][j] = acc[2][j] = acc[3][j] = acc[0][j];
            }
        }
        bf16x8 fb_keep[8];  // (timing-only builds with B16_ABL & 8: the weight fragments of the tile's first K step, multiplied again and again)
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            // stage kt of this tile is in LDS: stage 0 was requested before the previous tile's epilogue (18 younger operations
            // may still be in flight), every later stage one K step ago with nothing behind it
            if (kt == 0 && !first) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // everybody's pieces of stage kt have landed; everybody is done with the other buffer
            if (kt + 1 < nk) issue_tile(a_rs, cur ^ 1, (B16_ABL & 1) ? 0x80000000u : (unsigned)(kt + 1) * (4u * BKT));
            const float *Ac = wlds + cur * STAGE + (wr * 64) * BKT;
            const float *Bc = wlds + cur * STAGE + TM * BKT + (wc * 128) * BKT;
#pragma unroll
            for (int kc = 0; kc < BKT / 16; ++kc) {
                bf16x8 fa[4], fb[8];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Ac[dswz<BKT>(i * 16 + r16, kc * 4 + q)]));
                if (!(B16_ABL & 8) || (kt == 0 && kc == 0)) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(&Bc[dswz<BKT>(j * 16 + r16, kc * 4 + q)]));
                    if (B16_ABL & 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) fb_keep[j] = fb[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) fb[j] = fb_keep[j];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        first = false;
        __syncthreads();  // every wave is done with the last stage: both buffers are free
        // ---- the next tile's first stage goes into buffer 0 now; this tile leaves through buffer 1
        const int next_tile = row_tile + row_stride;
        const bool more = next_tile < row_tiles;
        if (more) {
            a_rs = a_rsrc(next_tile);
            issue_tile(a_rs, 0, 0);
        } else {  // keep the count of operations behind the (absent) stage uniform: nothing waits on it again
        }
        if (DX) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                mask_next[h] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bits_rs, more ? bits_off(next_tile, h) : 0xFFFFFFF0u, 0, 0));
        }
        const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(reinterpret_cast<char *>(Cb) + m0 * ldcb_b + (int64_t)(n0 + wc * 128) * 2,
                                                      (B16_ABL & 16) ? 0u : (unsigned)(rows_here - 1) * ldcb_b + 128 * 2);
        char *mine = reinterpret_cast<char *>(wlds + STAGE) + wave_u * (32 * 256);  // 8 KiB of buffer 1
        const int rr = lane >> 4, c16 = lane & 15;
        if (B16_ABL & 32) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // one rounding to bf16; forward: relu + bitmask first, dX: the forward's mask afterwards.  Written for the vector ALU: this
            // used to be ~1500 instructions per lane and tile (two v_max, a compare + select + shift + or per bit, one conversion per
            // element) = 5 us of a 20 us tile with the MFMA pipe idle; now relu is one v_max, a bit is v_sub (0 - r: sign set iff r > 0,
            // +0 stays +0) + v_alignbit (shift the word left, take that sign), a PAIR converts with one v_cvt_pk_bf16_f32.
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            u32x2 pk[2][8];
            if (!DX) {
                unsigned word[2];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    unsigned w = 0;
#pragma unroll
                    for (int j = 7; j >= 0; --j) {  // descending: bit (ii * 8 + j) * 4 + e ends up at its own position
                        float r[4];
#pragma unroll
                        for (int e = 3; e >= 0; --e) {
                            const float x = acc[2 * h + ii][j][e];
                            asm volatile("v_max_f32 %0, 0, %1" : "=v"(r[e]) : "v"(x));  // (the MFMAs are long complete: a barrier ago)
                            w = __builtin_amdgcn_alignbit(w, __float_as_uint(0.f - r[e]), 31);
                        }
                        pk[ii][j] = u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r[0], r[1]}, bf16x2)),
                                          __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r[2], r[3]}, bf16x2))};
                    }
                    word[ii] = w;
                }
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{word[0], word[1]}, bits_rs, bits_off(row_tile, h), 0, 0);
            } else {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int w = (int)mask_word[h][ii];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float r[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = acc[2 * h + ii][j][e];
                            const unsigned m = (unsigned)__builtin_amdgcn_sbfe(w, j * 4 + e, 1);  // 0 or ~0
                            r[e] = __uint_as_float(__float_as_uint(x) & m);  // masking before the rounding: a masked 0 stays +0
                        }
                        pk[ii][j] = u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r[0], r[1]}, bf16x2)),
                                          __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r[2], r[3]}, bf16x2))};
                    }
                }
            }
            // park the wave's 32 x 128 half (16-byte chunk c of row r at chunk c ^ (r & 15)), read it back as rows: 16 bytes per
            // lane, 4 rows x 256 contiguous bytes per wave-instruction (the wave reads only what it wrote: no barrier)
            if (h == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the first half's read-back is in registers
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 16 * ii + r16, ch = (2 * j + (q >> 1)) ^ (r & 15);
                    *reinterpret_cast<u32x2 *>(mine + r * 256 + ch * 16 + 8 * (q & 1)) = pk[ii][j];
                }
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const unsigned o_off = (unsigned)(wr * 64 + 32 * h + rr) * ldcb_b + c16 * 16;
#pragma unroll
            for (int g = 0; g < 2; ++g) {  // (four rows of registers at a time: the accumulators of the other half are still live)
                u32x4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = 4 * (4 * g + t) + rr;
                    v[t] = *reinterpret_cast<const u32x4 *>(mine + r * 256 + ((c16 ^ (r & 15)) * 16));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) __builtin_amdgcn_raw_buffer_store_b128(v[t], o_rs, o_off, 4 * (4 * g + t) * ldcb_b, 0);
            }
        }
    }
}

// Applicability of the bf16-in-memory forward: K a multiple of 64 (one LDS tile row = 64 k-values), hidden widths a multiple
// of 128 (the bitmask tile geometry).  Everything else takes the fp32 kernels on the ROUNDED fp32 copies -- the same products
// (a product of two bf16 values is exact in fp32), only slower -- followed by round_rows (optim.hip).
static int g_b16_wide = 2;  // rlppo_dbg_set(23, .): hidden / dX products of the bf16 update precision on 128 x 128 tiles (0), 256 x 256 tiles,
                            // one workgroup per tile (1), or 256 x 256 tiles walked by persistent workgroups (2, default [r3])
void set_b16_wide_tiles(int on) { g_b16_wide = on < 0 ? 0 : (on > 2 ? 2 : on); }
bool nt_b16_ok(int N, int K, bool hidden) {
    if (K % 64 != 0) return false;
    return hidden ? N % 128 == 0 : (N % 128 == 0 || N == 96 || N == 64 || N == 32);
}

// mode: 0 = output layer, 1 = hidden layer, 2 = masked + rounded dX (bias unused, bits read, Cb only)
int launch_gemm_nt_b16(hipStream_t st, const unsigned short *A, int64_t lda, const unsigned short *B, int64_t ldb, const float *bias,
                       float *C, int64_t ldc, unsigned short *Cb, int64_t ldcb, int64_t M, int N, int K, int epi, int mode,
                       unsigned long long *bits) {
    if (M <= 0) return 0;
    const bool hidden = mode != B16_OUT;
    RLPPO_CHECK_ARG(mode >= B16_OUT && mode <= B16_DX && nt_b16_ok(N, K, hidden) && A && B && (bias || mode == B16_DX),
                    "gemm_nt (bf16 in memory): N=%d K=%d mode=%d not supported", N, K, mode);
    if (mode == B16_DX) {
        RLPPO_CHECK_ARG(epi == EPI_MASK && bits && Cb && !C, "gemm_nt (bf16 in memory): the dX form needs the bitmask and a bf16 output");
        ldc = 0;
    }
    RLPPO_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && (!C || (ldc % 4 == 0 && ldc >= N)) &&
                        (!Cb || (ldcb % 4 == 0 && ldcb >= N)) && (C || Cb),
                    "gemm_nt (bf16 in memory): leading dimensions lda=%ld ldb=%ld ldc=%ld ldcb=%ld", (long)lda, (long)ldb, (long)ldc,
                    (long)ldcb);
    const int64_t lim = (int64_t)1 << 31;
    RLPPO_CHECK_ARG(129 * lda * 2 < lim && 129 * ldb * 2 < lim && 129 * ldc * 4 < lim && 129 * ldcb * 2 < lim,
                    "gemm_nt (bf16 in memory): a leading dimension is too wide for 32-bit tile offsets");
    const unsigned la = (unsigned)(lda * 2), lb = (unsigned)(ldb * 2), lc = (unsigned)(ldc * 4), lcb = (unsigned)(ldcb * 2);
    if (mode != B16_OUT && N % 256 == 0 && M >= 1024 && Cb && g_b16_wide && 257 * lda * 2 < lim && 257 * ldb * 2 < lim &&
        257 * ldc * 4 < lim && 257 * ldcb * 2 < lim) {  // 256 x 256 tiles
        static PerDeviceOnce attr_set[2];
        const int which = mode == B16_DX ? 1 : 0;
        const int rt128 = (int)cdiv(M, 128);
        if (g_b16_wide == 2 && !C && bits && N / 256 <= 8) {  // [r3] persistent workgroups (no fp32 copy of the output in this form)
            static PerDeviceOnce pattr_set[2];
            constexpr int LDS_BYTES = 2 * (256 + 256) * 32 * 4 + 1024;  // two stages + the column tile's bias
            int cus = 0;
            if (int rc_ = device_cu_count(&cus)) return rc_;
            const int col_tiles = N / 256, row_tiles = (int)cdiv(M, 256), group = 8 * col_tiles;
            int grid = cus / group * group;  // whole groups of 8 row tiles x all column tiles; never more workgroups than CUs
            const int need = (int)cdiv(row_tiles, 8) * group;
            grid = grid < group ? group : (grid > need ? need : grid);
#define B16P(MODE_)                                                                                                          \
    do {                                                                                                                     \
        if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_b16p_kernel<MODE_>, LDS_BYTES, pattr_set[which])) return rc_; \
        hipLaunchKernelGGL((gemm_nt_b16p_kernel<MODE_>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st, A, la, B, lb, bias, Cb, lcb, M, \
                           K, bits, row_tiles, col_tiles, rt128);                                                            \
    } while (0)
            if (which) B16P(B16_DX);
            else B16P(B16_HIDDEN);
#undef B16P
            RLPPO_LAUNCH_CHECK();
            return 0;
        }
#define B16W(MODE_)                                                                                                          \
    do {                                                                                                                     \
        constexpr int LDS_BYTES = 2 * (256 + 256) * 32 * 4; /* two 64 KiB stages; the epilogue parks 16 KiB per wave in them */ \
        if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_b16w_kernel<MODE_, 256, 32, 2>, LDS_BYTES, attr_set[which])) return rc_; \
        hipLaunchKernelGGL((gemm_nt_b16w_kernel<MODE_, 256, 32, 2>), dim3((unsigned)cdiv(M, 256), (unsigned)(N / 256)),        \
                           dim3(512), LDS_BYTES, st, A, la, B, lb, bias, C, lc, Cb, lcb, M, K, bits, rt128);                 \
    } while (0)
        if (which) B16W(B16_DX);
        else B16W(B16_HIDDEN);
#undef B16W
        RLPPO_LAUNCH_CHECK();
        return 0;
    }
    const int nb = N % 128 == 0 ? 8 : N / 16;
    dim3 grid((unsigned)cdiv(M, SBM), (unsigned)(N / (nb * 16)));
#define B16(NBV, E, H)                                                                                                      \
    hipLaunchKernelGGL((gemm_nt_b16_kernel<NBV, E, H>), grid, dim3(256), 0, st, A, la, B, lb, bias, C, lc, Cb, lcb, M, K, bits)
    if (mode == B16_DX) {
        B16(8, EPI_MASK, B16_DX);
    } else if (hidden) {
        RLPPO_CHECK_ARG(epi == EPI_BIAS_RELU, "gemm_nt (bf16 in memory): hidden layers are bias + ReLU");
        B16(8, EPI_BIAS_RELU, B16_HIDDEN);
    } else {
        RLPPO_CHECK_ARG((epi == EPI_BIAS || epi == EPI_BIAS_TANH) && C, "gemm_nt (bf16 in memory): output layer epilogue %d", epi);
        const bool th = epi == EPI_BIAS_TANH;
        switch (nb) {
            case 8: if (th) B16(8, EPI_BIAS_TANH, B16_OUT); else B16(8, EPI_BIAS, B16_OUT); break;
            case 6: if (th) B16(6, EPI_BIAS_TANH, B16_OUT); else B16(6, EPI_BIAS, B16_OUT); break;
            case 4: if (th) B16(4, EPI_BIAS_TANH, B16_OUT); else B16(4, EPI_BIAS, B16_OUT); break;
            default: if (th) B16(2, EPI_BIAS_TANH, B16_OUT); else B16(2, EPI_BIAS, B16_OUT); break;
        }
    }
#undef B16
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ gemm_tn, bf16 in memory
// Weight-gradient product of the bf16 update precision: dW[n][k] = sum_m dY[m][n] X[m][k] with BOTH operands bf16 in memory
// (dY: the rounded, masked gradient the dX kernel left; X: the rounded activation the forward left) on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation, the same partial-tile output as gemm_tn_dma_kernel (so tn_reduce_kernel, the
// fixed summation order and the bias column sums are shared).
// The contraction runs over ROWS of both operands, so an MFMA operand -- 8 consecutive m for one column -- is strided in
// memory.  The stage (64 rows x 128 columns of each operand, 256-byte rows) goes global -> LDS by LDS-DMA exactly as it lies
// in memory and is read with ds_read_b64_tr_b16, gfx950's transposing LDS read: per 16-lane group it takes a 4-row x
// 16-column block and hands lane i column i (cdna_hip_programming.md T10).  Two reads (rows 8q..8q+3, 8q+4..8q+7) make one
// operand.  LDS image: 16-byte chunk c of row r sits at chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3)) of the row (T10 image
// (b): the 4 rows of a block land 16 banks apart, the two blocks of a 32-lane half 8 banks apart -- conflict-free), applied
// on the DMA's SOURCE address.  The bias gradient is one more MFMA per n block against a constant operand of ones.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr16(const char *lds_base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3))) *)(const __attribute__((address_space(3))) char *)(lds_base + byte_off));
}
constexpr int TNB_ROWS = 64;  // rows per stage
__global__ __launch_bounds__(256, 2) void gemm_tn_b16_kernel(const unsigned short *__restrict__ dY, unsigned ldy_b,
                                                             const unsigned short *__restrict__ X, unsigned ldx_b, bool with_db,
                                                             int out, int64_t M, int rows_per_wg, float *__restrict__ partial) {
    constexpr int TMT = TNB_ROWS, TILE_B = TMT * 256;  // bytes of one operand's stage
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_B];  // [2 buffers][Y | X]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;  // XCD-aware order: see gemm_tn_dma_kernel
    {
        const int T = gridDim.x * gridDim.y;
        if ((gridDim.z & 7) == 0 && T > 1) {
            const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, g = id % (8 * T);
            const int tile = g >> 3;
            bz = (id / (8 * T)) * 8 + (g & 7);
            bx = tile % gridDim.x;
            by = tile / gridDim.x;
        }
    }
    const int n0 = bx * 128, k0 = by * 128;
    const int64_t mbeg = (int64_t)bz * rows_per_wg;
    const int rows = (int)((M - mbeg) < rows_per_wg ? (M - mbeg) : rows_per_wg);  // >= 1
    const int steps = (rows + TMT - 1) / TMT;
    const int rem = rows - (steps - 1) * TMT;  // rows of the last stage, 1..TMT

    // DMA: waves 0,1 stage dY (rows 0..31 / 32..63 of the stage), waves 2,3 stage X; one instruction = 4 rows of 256 bytes
    const bool is_x = wave_u >= 2;
    const unsigned ld_b = is_x ? ldx_b : ldy_b;
    const char *src = is_x ? reinterpret_cast<const char *>(X) + mbeg * ldx_b + (int64_t)k0 * 2
                           : reinterpret_cast<const char *>(dY) + mbeg * ldy_b + (int64_t)n0 * 2;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (unsigned)(rows - 1) * ld_b + 256u);
    const int half = wave_u & 1;
    unsigned goff[4];  // per-lane source offset of row group g & 3 (the swizzle depends on (row >> 2) & 3)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = 32 * half + 4 * g + (lane >> 4);
        const int lch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (g & 3));  // (row & 3) == lane >> 4, ((row >> 2) & 3) == g & 3
        goff[g] = (unsigned)row * ld_b + lch * 16;
    }
    auto issue_stage = [&](int buf, int stage) {
        char *dst = lds + (2 * buf + (is_x ? 1 : 0)) * TILE_B + half * 32 * 256;
        const unsigned sbase = (unsigned)stage * TMT * ld_b;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, reinterpret_cast<float *>(dst + g * 1024), 16, goff[g & 3],
                                                     sbase + (g >> 2) * 16 * ld_b, 0, 0);
    };
    // a ragged last stage: the DMA drops the rows past the split, so their (stale) LDS rows are cleared by hand
    auto clear_tail = [&](int buf) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = rem + (tid >> 4); r < TMT; r += 16) {
            *reinterpret_cast<f32x4 *>(lds + (2 * buf) * TILE_B + r * 256 + (tid & 15) * 16) = z;
            *reinterpret_cast<f32x4 *>(lds + (2 * buf + 1) * TILE_B + r * 256 + (tid & 15) * 16) = z;
        }
    };

    f32x4 acc[4][4], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool want_db = with_db && by == 0 && (wave_u & 1) == 0;  // scalar condition: waves with wk == 0

    // transposed-read addresses (bytes inside an operand's stage) of column block cb, half-octet h, for the 8-row octet q of
    // MFMA step 0; MFMA step s adds 32 rows (the swizzle only looks at row bits 0..3, which 32 s leaves alone)
    int ay[4][2], ax[4][2];
    {
        const int qq = r16 >> 2, p = r16 & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * q + 4 * h + qq;
                const int sw = (qq << 2) | ((2 * q + h) & 3);
                ay[c][h] = row * 256 + (((2 * (wn * 4 + c) + (p >> 1)) ^ sw) << 4) + 8 * (p & 1);
                ax[c][h] = row * 256 + (((2 * (wk * 4 + c) + (p >> 1)) ^ sw) << 4) + 8 * (p & 1);
            }
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (steps == 1 && rem < TMT) clear_tail(0);
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        const bool more = (st + 1) < steps;
        if (more) issue_stage(cur ^ 1, st + 1);
        const char *Yc = lds + (2 * cur) * TILE_B;
        const char *Xc = lds + (2 * cur + 1) * TILE_B;
#pragma unroll
        for (int s = 0; s < TMT / 32; ++s) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const s16x4 a0 = lds_tr16(Yc, ay[c][0] + s * 32 * 256), a1 = lds_tr16(Yc, ay[c][1] + s * 32 * 256);
                const s16x4 b0 = lds_tr16(Xc, ax[c][0] + s * 32 * 256), b1 = lds_tr16(Xc, ax[c][1] + s * 32 * 256);
                fa[c] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                fb[c] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (want_db) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above the wait: they are what hides the DMA latency
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more && (st + 2) == steps && rem < TMT) clear_tail(cur ^ 1);
        __syncthreads();
    }

    if (want_db && r16 == 0) {  // every column of the ones product holds the same sum: lanes with k = 0 write it
        float *pdb = partial + (size_t)gridDim.z * gridDim.y * gridDim.x * (128 * 128) + ((size_t)bz * gridDim.x + bx) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int nl = wn * 64 + i * 16 + 4 * q + e;
                if (n0 + nl < out) pdb[nl] = accb[i][e];
            }
    }
    {  // partial tile, the layout tn_reduce_kernel reads; last instructions of the wave (store-data hazard rule)
        const size_t tile_id = (size_t)bz * (gridDim.x * gridDim.y) + (size_t)by * gridDim.x + bx;
        const __amdgpu_buffer_rsrc_t p_rs = make_rsrc(partial + tile_id * (128 * 128), 128 * 128 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) stb(p_rs, (unsigned)tid * 16, (unsigned)(i * 4 + j) * 4096, acc[i][j]);
    }
}

// The same product on a 256 x 256 output tile: 8 waves as 4 (n) x 2 (k), each 64 x 128.  The 128 x 128 form brings 32 KiB into LDS
// per 64-row stage for 32 MFMAs per wave and waits for memory 62 % of its cycles (SQ_WAIT_ANY): it is bound by the tile bytes the
// LDS can keep in flight; a 256 x 256 tile needs half the bytes per flop.  A stage is four images of the 128 x 128 form (dY columns
// 0-127 / 128-255, X columns 0-127 / 128-255, each 64 rows x 256 bytes with the same swizzle), two stages = 128 KiB of dynamic LDS,
// one workgroup per CU.  The partial tiles keep the 128 x 128 layout (a wave writes into two tiles), so the reduction is shared.
__global__ __launch_bounds__(512, 1) void gemm_tn_b16w_kernel(const unsigned short *__restrict__ dY, unsigned ldy_b,
                                                              const unsigned short *__restrict__ X, unsigned ldx_b, bool with_db,
                                                              int out, int64_t M, int rows_per_wg, float *__restrict__ partial) {
    constexpr int TMT = TNB_ROWS, IMG_B = TMT * 256, STAGE_B = 4 * IMG_B;  // bytes
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    char *lds = reinterpret_cast<char *>(wlds);  // [2 stages][dY lo | dY hi | X lo | X hi]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wr = wave_u >> 1, wc = wave_u & 1;  // n rows 64 wr .., k columns 128 wc ..
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;  // XCD-aware order: see gemm_tn_dma_kernel
    {
        const int T = gridDim.x * gridDim.y;
        if ((gridDim.z & 7) == 0 && T > 1) {
            const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, g = id % (8 * T);
            const int tile = g >> 3;
            bz = (id / (8 * T)) * 8 + (g & 7);
            bx = tile % gridDim.x;
            by = tile / gridDim.x;
        }
    }
    const int n0 = bx * 256, k0 = by * 256;
    const int64_t mbeg = (int64_t)bz * rows_per_wg;
    const int rows = (int)((M - mbeg) < rows_per_wg ? (M - mbeg) : rows_per_wg);  // >= 1
    const int steps = (rows + TMT - 1) / TMT;
    const int rem = rows - (steps - 1) * TMT;

    // DMA: wave w stages image w >> 1 (0, 1: dY column halves; 2, 3: X column halves), rows 32 (w & 1) .. + 31 of the stage
    const int img = wave_u >> 1, half = wave_u & 1;
    const bool is_x = img >= 2;
    const unsigned ld_b = is_x ? ldx_b : ldy_b;
    const char *src = (is_x ? reinterpret_cast<const char *>(X) + mbeg * ldx_b + (int64_t)k0 * 2
                            : reinterpret_cast<const char *>(dY) + mbeg * ldy_b + (int64_t)n0 * 2) + (img & 1) * 256;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (unsigned)(rows - 1) * ld_b + 256u);
    unsigned goff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = 32 * half + 4 * g + (lane >> 4);
        const int lch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (g & 3));
        goff[g] = (unsigned)row * ld_b + lch * 16;
    }
    auto issue_stage = [&](int buf, int stage) {
        char *dst = lds + buf * STAGE_B + img * IMG_B + half * 32 * 256;
        const unsigned sbase = (unsigned)stage * TMT * ld_b;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, reinterpret_cast<float *>(dst + g * 1024), 16, goff[g & 3],
                                                     sbase + (g >> 2) * 16 * ld_b, 0, 0);
    };
    auto clear_tail = [&](int buf) {  // rows past a ragged split: the DMA dropped them, clear the stale LDS rows (all four images)
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = rem + (tid >> 6); r < TMT; r += 8)
            *reinterpret_cast<f32x4 *>(lds + buf * STAGE_B + ((tid >> 4) & 3) * IMG_B + r * 256 + (tid & 15) * 16) = z;
    };

    f32x4 acc[4][8], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool want_db = with_db && by == 0 && wc == 0;  // scalar condition

    int ay[4][2], ax[8][2];  // transposed-read addresses inside an image: see gemm_tn_b16_kernel
    {
        const int qq = r16 >> 2, p = r16 & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * q + 4 * h + qq;
            const int sw = (qq << 2) | ((2 * q + h) & 3);
#pragma unroll
            for (int c = 0; c < 4; ++c) ay[c][h] = row * 256 + (((2 * ((wr & 1) * 4 + c) + (p >> 1)) ^ sw) << 4) + 8 * (p & 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) ax[c][h] = row * 256 + (((2 * c + (p >> 1)) ^ sw) << 4) + 8 * (p & 1);
        }
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (steps == 1 && rem < TMT) clear_tail(0);
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        const bool more = (st + 1) < steps;
        if (more) issue_stage(cur ^ 1, st + 1);
        const char *Yc = lds + cur * STAGE_B + (wr >> 1) * IMG_B;
        const char *Xc = lds + cur * STAGE_B + (2 + wc) * IMG_B;
#pragma unroll
        for (int s = 0; s < TMT / 32; ++s) {
            bf16x8 fa[4], fb[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const s16x4 a0 = lds_tr16(Yc, ay[c][0] + s * 32 * 256), a1 = lds_tr16(Yc, ay[c][1] + s * 32 * 256);
                fa[c] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const s16x4 b0 = lds_tr16(Xc, ax[c][0] + s * 32 * 256), b1 = lds_tr16(Xc, ax[c][1] + s * 32 * 256);
                fb[c] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (want_db) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more && (st + 2) == steps && rem < TMT) clear_tail(cur ^ 1);
        __syncthreads();
    }

    // the 128 x 128-tile layout tn_reduce_kernel reads: this wave's 64 n rows are quadrant row (wr & 1) of n tile 2 bx + (wr >> 1), its
    // 128 k columns are both quadrant columns (j >> 2) of k tile 2 by + wc
    const int tx128 = gridDim.x * 2, ty128 = gridDim.y * 2;
    const int nt = bx * 2 + (wr >> 1), kt = by * 2 + wc, wn = wr & 1;
    if (want_db && r16 == 0) {
        float *pdb = partial + (size_t)gridDim.z * ty128 * tx128 * (128 * 128) + ((size_t)bz * tx128 + nt) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int nl = wn * 64 + i * 16 + 4 * q + e;
                if (nt * 128 + nl < out) pdb[nl] = accb[i][e];
            }
    }
    {
        const size_t tile_id = (size_t)bz * (tx128 * ty128) + (size_t)kt * tx128 + nt;
        const __amdgpu_buffer_rsrc_t p_rs = make_rsrc(partial + tile_id * (128 * 128), 128 * 128 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                stb(p_rs, (unsigned)((wn * 2 + (j >> 2)) * 64 + lane) * 16, (unsigned)(i * 4 + (j & 3)) * 4096, acc[i][j]);
    }
}

// rows per workgroup of the 256 x 256-tile bf16 form: one round of one workgroup per CU
int64_t tn_partial_rows_wide(int pout, int pin, int64_t M) {
    const int64_t tiles = (int64_t)(pout / 256) * (pin / 256);
    const int64_t rows = round_up(cdiv(M * tiles, 256), 64);
    return rows > 256 ? rows : 256;
}

// The bf16-in-memory form of launch_gemm_tn: padded widths pout / pin multiples of 128 (whole tiles; the operands' columns
// beyond out / in are zero padding), leading dimensions in elements; dW[out][in], db[out] are the unpadded gradient arrays.
bool tn_b16_ok(int pout, int pin) { return pout > 0 && pin > 0 && pout % 128 == 0 && pin % 128 == 0; }
int launch_gemm_tn_b16(hipStream_t st, const unsigned short *dY, int64_t ldy, const unsigned short *X, int64_t ldx, float *dW,
                       float *db, int pout, int pin, int out, int in, int64_t M, float *ws, size_t ws_floats) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(tn_b16_ok(pout, pin) && ldy % 8 == 0 && ldx % 8 == 0 && pout <= ldy && pin <= ldx && out <= pout && in <= pin &&
                        out > pout - 128 && in > pin - 128 && dY && X && dW,
                    "gemm_tn (bf16 in memory): bad shapes pout=%d pin=%d out=%d in=%d ldy=%ld ldx=%ld", pout, pin, out, in, (long)ldy,
                    (long)ldx);
    if (!ws || ws_floats < tn_partial_floats(out, in, M)) {
        set_error("gemm_tn (bf16 in memory): workspace %zu < %zu floats", ws ? ws_floats : (size_t)0, tn_partial_floats(out, in, M));
        return RLPPO_ERR_WORKSPACE;
    }
    const int rows_per_wg = (int)round_up(tn_partial_rows(out, in, M), TNB_ROWS);
    const int64_t lim = (int64_t)1 << 30;
    RLPPO_CHECK_ARG((rows_per_wg + TNB_ROWS) * ldy * 2 < lim && (rows_per_wg + TNB_ROWS) * ldx * 2 < lim,
                    "gemm_tn (bf16 in memory): a leading dimension is too wide for 32-bit tile offsets");
    const int tiles_x = pout / 128, tiles_y = pin / 128;
    int splits = (int)cdiv(M, rows_per_wg);
    if (pout % 256 == 0 && pin % 256 == 0 && g_b16_wide) {  // 256 x 256 tiles (same partial-tile layout, its own split count)
        const int rows_w = (int)tn_partial_rows_wide(pout, pin, M);
        RLPPO_CHECK_ARG((rows_w + TNB_ROWS) * ldy * 2 < lim && (rows_w + TNB_ROWS) * ldx * 2 < lim,
                        "gemm_tn (bf16 in memory): a leading dimension is too wide for 32-bit tile offsets");
        splits = (int)cdiv(M, rows_w);
        static PerDeviceOnce attr_set;
        constexpr int LDS_BYTES = 2 * 4 * TNB_ROWS * 256;
        if (int rc_ = set_dynamic_lds_once((const void *)gemm_tn_b16w_kernel, LDS_BYTES, attr_set)) return rc_;
        hipLaunchKernelGGL(gemm_tn_b16w_kernel, dim3((unsigned)(pout / 256), (unsigned)(pin / 256), (unsigned)splits), dim3(512), LDS_BYTES,
                           st, dY, (unsigned)(ldy * 2), X, (unsigned)(ldx * 2), db != nullptr, out, M, rows_w, ws);
    } else {
        dim3 grid((unsigned)tiles_x, (unsigned)tiles_y, (unsigned)splits);
        hipLaunchKernelGGL(gemm_tn_b16_kernel, grid, dim3(256), 0, st, dY, (unsigned)(ldy * 2), X, (unsigned)(ldx * 2), db != nullptr,
                           out, M, rows_per_wg, ws);
    }
    RLPPO_LAUNCH_CHECK();
    return launch_tn_reduce(st, ws, splits, tiles_x, tiles_y, dW, db, out, in);
}
}  // namespace rlppo
