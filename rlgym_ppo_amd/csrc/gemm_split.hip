// gemm_split.hip -- [r4, OPT-IN: rlppo_set_update_precision(2)] the hidden-layer forward and dX products of the PPO update with fp32
// data in memory and fp32-grade results, computed on the bf16 MFMA pipe.
//
// gfx950's fp32-input MFMA runs at 1/16 of its bf16 rate (157 vs 2500 TFLOP/s), and the fp32 update sits at 0.78 of that peak.  A
// product that must stay fp32-accurate can still use the fast pipe: split every fp32 operand into three bf16 pieces,
//     x = x_h + x_m + x_l     (x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m): |x_m| <= 2^-9 |x|, |x_l| <= 2^-18 |x|),
// and keep the six piece products whose index sum is <= 2:
//     x . w  ~=  x_h.w_h + (x_h.w_m + x_m.w_h + x_m.w_m + x_h.w_l + x_l.w_h)           (dropped: <= 2^-26 |x||w|)
// Each is one v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The five small products of a 32-wide K block are summed among
// THEMSELVES first (a chain that starts from 0: its roundings are 2^-8 of the large product's) and enter the accumulator with one
// addition, so the accumulator sees ONE MFMA and one add per K block: measured against float64 this is MORE accurate than the
// fp32 MFMA's fmaf chain (0.43-0.46 x its error at K = 256, profiles/r04_split_bf16_probe.txt) at 1.35 x its speed.
//
// Operands: A [M][K] fp32 exactly as the update stores its activations / activation gradients -- split ON THE FLY, per MFMA
// fragment, in registers (v_cvt_pk_bf16_f32 + shift / mask + subtract: 5.5 vector instructions per value); the weights pre-split
// by rlppo_net_pack_x3 after every optimiser step into three bf16 planes in STAGE-MAJOR order [K / 32][3][N / 16][1 KiB block], so
// that the 48 KiB a K step needs are contiguous in memory, every 16-column block in the chunk order wpos() that makes its fragment
// reads bank-conflict-free.  One 256 x 256 output tile per workgroup and pass, 8 waves, 128 accumulator registers per lane, one
// workgroup per CU, two stages of 80 KiB: A 256 x 32 fp32 (the 128-byte-row image and swizzle of the bf16 kernels) + three 16 KiB
// weight planes.  Two kernels: gemm_nt_split_kernel, one tile per workgroup (waves 4 x 2, small launches, any K % 32 == 0), and the
// persistent gemm_nt_split_p_kernel (a wave owns 32 rows x 256 columns and splits step s + 1 under the MFMAs of step s).  The tile
// leaves through LDS in row pieces.  FWD: C = relu(A . W^T + b) and the ReLU bitmask in the layout every other kernel of the update
// reads (128 x 128 tiles, one 64-bit word per lane); DX: C = (A . B^T) masked by the bitmask of the layer below.
//
// Where a launch's time goes (M = 524,288, 256 -> 256; profiles/r04_split_ablation.txt, PMC in the same file): 1.07 GB of HBM traffic
// (floor ~230 us at the rate a copy reaches) against 166 us of MFMA-pipe time (25.2 M MFMAs x 16 cycles / 1,024 SIMDs at 2.36 GHz);
// the launch takes 345-375 us = 0.48 MFMA-busy.  The K loop alone (no traffic, no epilogue, no barrier) runs 219 us: at 16 cycles
// per MFMA, 8 of which hold the SIMD's vector issue, a wave-step's 218 vector instructions, 52 fragment reads and 10 stage requests
// (60+ cycles of issue each) do not fit the gaps of its 192 MFMAs; the epilogue's vector work adds 23 us, barriers and request
// issue 18 us, and the memory system 70-110 us that the loop does not hide (stores 32 us: every wave's 32 tile stores must be
// acknowledged before its next counted wait can pass, vmcnt being in order).
#include "gemm_detail.hpp"
// -DSPLIT_ABL=bits builds TIMING-ONLY variants of the persistent kernel (results are garbage) for tools/split_ablation.sh, which
// prices the parts of a K step by leaving them out: 1 the steps' stage requests go outside their descriptors (no traffic, still
// issued and counted), 2 no split of the next step's fragments, 8 only the large product, 16 the tile's stores dropped, 32 no epilogue
// work, 64 no stage-request instructions at all, 128 no barrier per step.  0 (the default) is the product.
#ifndef SPLIT_ABL
#define SPLIT_ABL 0
#endif

namespace rlppo {
namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// three bf16x8 MFMA fragments (8 consecutive k of one row) from 8 fp32 values, pieces rounded to nearest even
__device__ __forceinline__ void split8(const f32x4 &lo, const f32x4 &hi, bf16x8 &ph, bf16x8 &pm, bf16x8 &pl) {
    u32x4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = p < 2 ? lo[2 * p] : hi[2 * p - 4], x1 = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
        const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);  // exact
        const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
        const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);  // exact
        h[p] = hp;
        m[p] = mp;
        l[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
    }
    ph = __builtin_bit_cast(bf16x8, h);
    pm = __builtin_bit_cast(bf16x8, m);
    pl = __builtin_bit_cast(bf16x8, l);
}

constexpr int TM = 256, TN = 256, BK = 32;   // tile, K step
constexpr int A_STAGE = TM * BK * 4;         // 32 KiB
constexpr int W_PLANE = TN * BK * 2;         // 16 KiB
constexpr int STAGE = A_STAGE + 3 * W_PLANE; // 80 KiB
enum { SPLIT_FWD = 0, SPLIT_DX = 1 };
// The image of a weight plane's 16-column block (1 KiB: 16 columns x 32 k) in memory and in LDS: the 16-byte chunk (column r16, k quarter
// q) at chunk wpos(r16, q).  A fragment read is one ds_read_b128 with lane = q * 16 + r16, served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS); with plain rows (chunk r16 * 4 + q) columns r16 and r16 + 4 k of a group meet
// on a bank -- 2-way conflicts, 8 cycles per read; rotating q by -(r16 / 4) gives every group 16 distinct chunk residues mod 16.
__host__ __device__ constexpr int wpos(int r16, int q) { return r16 * 4 + (q ^ ((0 - (r16 >> 2)) & 3)); }

// planes: the N_total x K weight matrix as [K / 32][3][N_total][32] bf16; this workgroup's column tile is rows n0 .. n0 + 255 of every
// [N_total][32] block.  grid = (row tiles, column tiles).
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt_split_kernel(const float *__restrict__ A, unsigned lda_b, const unsigned short *__restrict__ planes,
                                                               int n_total, const float *__restrict__ bias, float *__restrict__ C, unsigned ldc_b,
                                                               int64_t M, int K, unsigned long long *__restrict__ bits, int row_tiles128) {
    constexpr bool DX = MODE == SPLIT_DX;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [2][A 256 x 128 B swizzled | 3 x W plane 256 x 64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wr = wave_u >> 1, wc = wave_u & 1;
    const int row_tile = blockIdx.x, col_tile = blockIdx.y;
    const int64_t m0 = (int64_t)row_tile * TM;
    const int n0 = col_tile * TN;
    const int rows_here = (int)((M - m0) < TM ? (M - m0) : TM);
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b, (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 4);
    // a K step's planes of this column tile: plane p at byte (kt * 3 + p) * n_total * 64 + n0 * 64, 16 KiB contiguous each
    const unsigned plane_stride = (unsigned)n_total * 64u;
    const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(reinterpret_cast<const char *>(planes) + (int64_t)n0 * 64, (unsigned)nk * 3u * plane_stride);
    // A pieces: 8 rows x 128 B per instruction (lane -> row lane / 8, chunk lane % 8, swizzled on the source side), 4 per wave and stage
    const int row_p = wave * 8 + lane / 8, pch = lane % 8, lch = pch ^ (row_p & 7);
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    auto issue = [&](int buf, int kt) {
        char *Ad = lds + buf * STAGE + wave_u * (8 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, reinterpret_cast<float *>(Ad + i * (64 * 128)), 16, a_off, (unsigned)kt * 128u + i * 64u * lda_b, 0, 0);
        // weights: 48 pieces of 1 KiB per stage, 6 per wave: piece g = plane g / 16, KiB g % 16 of it
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = wave_u * 6 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, reinterpret_cast<float *>(lds + buf * STAGE + A_STAGE + g * 1024), 16, (unsigned)lane * 16u,
                                                     ((unsigned)kt * 3u + (unsigned)(g >> 4)) * plane_stride + (unsigned)(g & 15) * 1024u, 0, 0);
        }
    };
    // bitmask words of this wave's two 32-row lane slots (128 x 128-tile layout of relu_bits, as gemm_nt_b16w_kernel)
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
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0][j] = *reinterpret_cast<const f32x4 *>(&bias[n0 + wc * 128 + j * 16 + q * 4]);
            acc[1][j] = acc[2][j] = acc[3][j] = acc[0][j];
        }
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
            split8(lo, hi, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            bf16x8 wh[4], wm[4], wl[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int col = wc * 128 + (jh * 4 + jj) * 16 + r16;
                const int at = (col & ~15) * 64 + wpos(r16, q) * 16;
                wh[jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + at));
                wm[jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + W_PLANE + at));
                wl[jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + 2 * W_PLANE + at));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], al[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[jj], ah[i], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], am[i], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], am[i], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[jj], ah[i], t, 0, 0, 0);
                    f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[jj], ah[i], acc[i][jh * 4 + jj], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[e] += t[e];
                    acc[i][jh * 4 + jj] = c;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: relu + bitmask (FWD) or the mask of the layer below (DX), in place
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned lo, hi;
        if (DX) {
            lo = (unsigned)mask_word[h];
            hi = (unsigned)(mask_word[h] >> 32);
        } else {
            lo = hi = 0;
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int b = (ii * 8 + j) * 4 + e;
                    if (DX) {
                        const unsigned m = 0u - (((b < 32 ? lo : hi) >> (b & 31)) & 1u);
                        acc[2 * h + ii][j][e] = __uint_as_float(__float_as_uint(acc[2 * h + ii][j][e]) & m);
                    } else {
                        const float x = relu1(acc[2 * h + ii][j][e]);
                        acc[2 * h + ii][j][e] = x;
                        const unsigned v = x > 0.f ? 1u : 0u;
                        if (b < 32) lo |= v << b;
                        else hi |= v << (b - 32);
                    }
                }
        if (!DX && bits && wlive[h]) bits[widx[h]] = ((unsigned long long)hi << 32) | lo;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the fp32 tile leaves through LDS, 16 rows of the wave's 64 x 128 sub-tile at a time: a lane holds 4 consecutive columns of
    // one row (a wave-instruction would store 16 rows x 64 bytes); parked in 8 KiB of the wave's own (16-byte chunk c of row r at
    // chunk c ^ r) and read back as rows it leaves as 2 rows x 512 contiguous bytes per wave-instruction.  In-wave: no barrier.
    __syncthreads();  // every wave is done with the last stage
    char *mine = lds + wave_u * (16 * 512);
    const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)(n0 + wc * 128) * 4, (unsigned)(rows_here - 1) * ldc_b + 128 * 4);
    const int rr = lane >> 5, c32 = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous quarter's read-back is in registers
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(mine + r16 * 512 + (((j * 4 + q) ^ r16) * 16)) = acc[i][j];
        f32x4 v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = 2 * t + rr;
            v[t] = *reinterpret_cast<const f32x4 *>(mine + r * 512 + ((c32 ^ r) * 16));
        }
        const unsigned c_off = (unsigned)(wr * 64 + i * 16 + rr) * ldc_b + c32 * 16;
#pragma unroll
        for (int t = 0; t < 8; ++t) stb(c_rs, c_off, 2 * t * ldc_b, v[t]);
    }
}

// ---- persistent form: ONE workgroup per CU walks the row tiles of its column tile (as gemm_nt_b16p_kernel, csrc/gemm_b16.hip), and
// a K step's split runs in the SHADOW of the previous step's MFMAs.
//
// Where the one-tile kernel's K step goes (profiles/r04_split_turn_taking_experiment.txt): 6 x 1060 cycles of MFMAs at the pipe's
// full rate and, before them, 2700 cycles in which no MFMA issues -- 8 fragment reads, the split of four A fragments (176 vector
// instructions per wave, every fragment split by the two waves that share its rows), the stage requests, the barrier.  Here a wave
// owns 32 rows x ALL 256 columns: no row is split twice (88 instructions per wave and step), its two A fragments fit a SECOND
// 24-register set, and the fragments of step s + 1 are read and split between the MFMAs of step s -- by the same wave, so both
// waves of a SIMD multiply at all times.  The price is LDS reads: every wave streams all three weight planes of the stage
// (48 ds_read_b128 per step, double-buffered one 16-column block ahead; 416 KiB per step and CU = a quarter of the step's LDS
// cycles now that the plane image is conflict-free, see wpos()).  The stage's two parts travel on their own schedules: step s
// requests W(s + 1) and A(s + 2), because A(s + 1) is consumed DURING step s; two A buffers suffice (A(s)'s was freed by step s - 1).
// The head of tile t + 1 -- W(0), A(0), A(1) -- is requested BEFORE the epilogue of tile t, which leaves through the other W buffer
// (4 KiB per wave, 16 rows x 64 columns at a time: 4 rows x 256 contiguous bytes per wave-instruction), with a counted wait (the head's
// 14 pieces are older than the epilogue's 2 bitmask accesses and 32 stores, which stay in flight).  Every step issues the same
// number of pieces: a stage that does not exist is requested outside its descriptor (dropped, still counted), because the compiler's
// own waits for register loads take the minimum over paths.  The forward adds its bias in the epilogue.  K / 32 must be even.
struct SplitFrags {
    bf16x8 h[2], m[2], l[2];
};
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hp, unsigned &mp, unsigned &lp) {
    hp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);  // exact
    mp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);  // exact
    lp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt_split_p_kernel(const float *__restrict__ A, unsigned lda_b, const unsigned short *__restrict__ planes,
                                                                 int n_total, const float *__restrict__ bias, float *__restrict__ C, unsigned ldc_b,
                                                                 int64_t M, int K, unsigned long long *__restrict__ bits, int row_tiles, int col_tiles,
                                                                 int row_tiles128) {
    constexpr bool DX = MODE == SPLIT_DX;
    constexpr unsigned DROPPED = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [2][A 256 x 128 B swizzled | 3 x W plane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // rows wave_u * 32 .. + 31 of the tile
    const int r16 = lane & 15, q = lane >> 4;
    const int b = blockIdx.x, group = 8 * col_tiles;
    const int col_tile = (b % group) >> 3;
    const int row_first = (b / group) * 8 + (b & 7), row_stride = (gridDim.x / group) * 8;
    const int n0 = col_tile * TN;
    const int nk = K / BK;
    if (row_first >= row_tiles) return;
    const unsigned plane_stride = (unsigned)n_total * 64u;
    const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(reinterpret_cast<const char *>(planes) + (int64_t)n0 * 64, (unsigned)nk * 3u * plane_stride);
    const __amdgpu_buffer_rsrc_t bits_rs = make_rsrc(bits, (unsigned)((size_t)row_tiles128 * (col_tiles * 2) * 256 * 8));
    const int row_p = wave * 8 + lane / 8, pch = lane % 8, lch = pch ^ (row_p & 7);
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    auto a_rsrc = [&](int row_tile) {
        const int64_t m0 = (int64_t)row_tile * TM;
        const int rows_here = (int)((M - m0) < TM ? (M - m0) : TM);
        return make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b, (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 4);
    };
    auto issue_a = [&](const __amdgpu_buffer_rsrc_t &a_rs, int buf, int kt, bool real) {  // 4 pieces per wave: 8 rows x 128 B each
        char *Ad = lds + buf * STAGE + wave_u * (8 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, reinterpret_cast<float *>(Ad + i * (64 * 128)), 16, a_off,
                                                     real ? (unsigned)kt * 128u + i * 64u * lda_b : DROPPED, 0, 0);
    };
    auto issue_w = [&](int buf, int kt, bool real) {  // 6 pieces per wave: KiB g % 16 of plane g / 16
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = wave_u * 6 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, reinterpret_cast<float *>(lds + buf * STAGE + A_STAGE + g * 1024), 16, (unsigned)lane * 16u,
                                                     real ? ((unsigned)kt * 3u + (unsigned)(g >> 4)) * plane_stride + (unsigned)(g & 15) * 1024u : DROPPED, 0, 0);
        }
    };
    // the bitmask word of this wave's 32 rows in column half hc (128 x 128-tile layout of relu_bits); outside the descriptor past the end
    auto bits_off = [&](int row_tile, int hc) -> unsigned {
        const int rt128 = row_tile * 2 + (wave_u >> 2), ct128 = col_tile * 2 + hc, nct128 = col_tiles * 2;
        return rt128 < row_tiles128 ? (unsigned)((((size_t)rt128 * nct128 + ct128) * 256 + (wave_u & 3) * 64 + lane) * 8) : 0xFFFFFFF0u;
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t a_rs = a_rsrc(row_first);
    issue_w(0, 0, true);
    issue_a(a_rs, 0, 0, true);
    issue_a(a_rs, 1, 1, true);
    bool first = true;
    u32x2 mask_next[2];
    if (DX) {
#pragma unroll
        for (int h = 0; h < 2; ++h) mask_next[h] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bits_rs, bits_off(row_first, h), 0, 0));
    }
    const int a_row0 = wave_u * 32 + r16;                     // fragment i: row a_row0 + 16 i (same swizzle key: 16 i does not touch row & 7)
    const int a_lo = a_row0 * 128 + (((2 * q) ^ (a_row0 & 7)) * 16), a_hi = a_row0 * 128 + (((2 * q + 1) ^ (a_row0 & 7)) * 16);
    const int w_frag = wpos(r16, q) * 16;                     // + plane * W_PLANE + block * 1024
    for (int row_tile = row_first; row_tile < row_tiles; row_tile += row_stride) {
        const int64_t m0 = (int64_t)row_tile * TM;
        const int rows_here = (int)((M - m0) < TM ? (M - m0) : TM);
        f32x4 acc[2][16];
        u32x2 mask_word[2];
        if (DX) {
#pragma unroll
            for (int h = 0; h < 2; ++h) mask_word[h] = mask_next[h];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the head is in LDS (its younger neighbours stay in flight: the previous epilogue's 2 bitmask accesses + 32 stores, or the
        // first tile's 2 mask loads of the DX form)
        if (!first) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DX ? 2 : 0) : "memory");
        __builtin_amdgcn_s_barrier();  // raw (no fence) throughout: a wave waits for ITS pieces before the barrier behind which they are read
        SplitFrags sp0, sp1;
        {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(lds + a_lo + i * (16 * 128)), hi = *reinterpret_cast<const f32x4 *>(lds + a_hi + i * (16 * 128));
                split8(lo, hi, sp0.h[i], sp0.m[i], sp0.l[i]);
            }
        }
        auto step = [&](const SplitFrags &cur, SplitFrags &nxt, int s) {
            if (s) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // W(s), A(s + 1): requested a step ago
            if (!(SPLIT_ABL & 128)) __builtin_amdgcn_s_barrier();    // everybody is done with W(s - 1) and with A(s)
            if (!(SPLIT_ABL & 64)) {
                issue_a(a_rs, s & 1, s + 2, !(SPLIT_ABL & 1) && s + 2 < nk);
                issue_w((s + 1) & 1, s + 1, !(SPLIT_ABL & 1) && s + 1 < nk);
            }
            const char *Wc = lds + (s & 1) * STAGE + A_STAGE + w_frag;
            const char *An = lds + ((s + 1) & 1) * STAGE;  // (the last step splits a stale stage into fragments nobody uses: no branch)
            f32x4 raw[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                raw[i][0] = *reinterpret_cast<const f32x4 *>(An + a_lo + i * (16 * 128));
                raw[i][1] = *reinterpret_cast<const f32x4 *>(An + a_hi + i * (16 * 128));
            }
            bf16x8 w[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) w[0][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + p * W_PLANE));
            u32x4 nh[2], nm[2], nl[2];
            f32x4 tp0, tp1;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j + 1 < 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        w[(j + 1) & 1][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Wc + p * W_PLANE + (j + 1) * 1024));
                }
                __builtin_amdgcn_sched_barrier(0);  // the reads stay AHEAD of this block's MFMAs (left alone the scheduler sinks them to the block's end)
                const bf16x8 wh = w[j & 1][0], wm = w[j & 1][1], wl = w[j & 1][2];
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                // the five small products of a K block summed among themselves, the two fragments' chains interleaved
                f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cur.l[0], z, 0, 0, 0);
                f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cur.l[1], z, 0, 0, 0);
                f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cur.h[0], acc[0][j], 0, 0, 0);
                f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cur.h[1], acc[1][j], 0, 0, 0);
                if (!(SPLIT_ABL & 8)) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, cur.h[0], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, cur.h[1], t1, 0, 0, 0);
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, cur.m[0], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, cur.m[1], t1, 0, 0, 0);
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cur.m[0], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, cur.m[1], t1, 0, 0, 0);
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, cur.h[0], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, cur.h[1], t1, 0, 0, 0);
                }
                acc[0][j] = c0;
                acc[1][j] = c1;
                if (j > 0) {  // the previous block's small sum enters its accumulator now: its chain has long finished
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0][j - 1][e] += tp0[e], acc[1][j - 1][e] += tp1[e];
                }
                tp0 = t0;
                tp1 = t1;
                if (j >= 2 && j < 10 && !(SPLIT_ABL & 2)) {  // one eighth of the next step's split: 2 values of fragment (j - 2) / 4
                    const int f = (j - 2) >> 2, pp = (j - 2) & 3;
                    const f32x4 src = raw[f][pp >> 1];
                    unsigned hp, mp, lp;
                    split2(src[2 * (pp & 1)], src[2 * (pp & 1) + 1], hp, mp, lp);
                    nh[f][pp] = hp, nm[f][pp] = mp, nl[f][pp] = lp;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][15][e] += tp0[e], acc[1][15][e] += tp1[e];
            if (SPLIT_ABL & 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) nh[i] = __builtin_bit_cast(u32x4, raw[i][0]), nm[i] = __builtin_bit_cast(u32x4, raw[i][1]), nl[i] = nh[i] ^ nm[i];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                nxt.h[i] = __builtin_bit_cast(bf16x8, nh[i]);
                nxt.m[i] = __builtin_bit_cast(bf16x8, nm[i]);
                nxt.l[i] = __builtin_bit_cast(bf16x8, nl[i]);
            }
        };
#pragma unroll 1
        for (int s = 0; s < nk; s += 2) {
            step(sp0, sp1, s);
            step(sp1, sp0, s + 1);
        }
        first = false;
        __syncthreads();  // every wave is done with the last stage (only dropped pieces are outstanding): both buffers are free
        // ---- forward: the bias, ahead of the next head in this wave's queue (the compiler's wait for it leaves the head in flight)
        f32x4 bias4[16];
        if (!DX) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) bias4[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(&bias[n0 + j * 16 + q * 4]));
            asm volatile("" ::: "memory");
        }
        // ---- the next tile's head goes into W buffer 0 and both A buffers now; this tile leaves through W buffer 1
        const int next_tile = row_tile + row_stride;
        const bool more = next_tile < row_tiles;
        if (more) a_rs = a_rsrc(next_tile);
        issue_w(0, 0, more);
        issue_a(a_rs, 0, 0, more);
        issue_a(a_rs, 1, 1, more);
        if (DX) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                mask_next[h] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bits_rs, more ? bits_off(next_tile, h) : 0xFFFFFFF0u, 0, 0));
        }
        if (!(SPLIT_ABL & 32))
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // column half h: bit (ii * 8 + j) * 4 + e = row block ii, column block j of the half, element e
            unsigned lo = DX ? mask_word[h][0] : 0u, hi = DX ? mask_word[h][1] : 0u;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int bb = (ii * 8 + j) * 4 + e;
                        if (DX) {
                            const unsigned m = 0u - (((bb < 32 ? lo : hi) >> (bb & 31)) & 1u);
                            acc[ii][h * 8 + j][e] = __uint_as_float(__float_as_uint(acc[ii][h * 8 + j][e]) & m);
                        } else {
                            const float x = relu1(acc[ii][h * 8 + j][e] + bias4[h * 8 + j][e]);
                            acc[ii][h * 8 + j][e] = x;
                            const unsigned v = x > 0.f ? 1u : 0u;
                            if (bb < 32) lo |= v << bb;
                            else hi |= v << (bb - 32);
                        }
                    }
            if (!DX) __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo, hi}, bits_rs, bits_off(row_tile, h), 0, 0);  // (dropped past the end, still counted)
        }
        __builtin_amdgcn_sched_barrier(0);
        // a lane holds 4 consecutive columns of one row: 16 rows x 64 columns are parked in 4 KiB of the wave's own (16-byte chunk c of
        // row r at chunk c ^ r) and read back as 4 rows x 256 contiguous bytes per wave-instruction.  In-wave: no barrier.
        int le = lane;
        asm volatile("" : "+v"(le));  // the addresses below are recomputed per tile: hoisted out of the tile loop they are spilled, and a
                                      // scratch reload is a vmcnt(0) wait in the middle of the epilogue
        const int er = le & 15, eq = le >> 4;
        char *mine = lds + STAGE + A_STAGE + wave_u * 4096;
        const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)n0 * 4, (SPLIT_ABL & 16) ? 0u : (unsigned)(rows_here - 1) * ldc_b + 256 * 4);
        if (!(SPLIT_ABL & 32))
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = it >> 2, cq = it & 3;
            if (it) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous piece's read-back is in registers
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4 *>(mine + er * 256 + (((jj * 4 + eq) ^ er) * 16)) = acc[i][cq * 4 + jj];
            f32x4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = 4 * t + eq;
                v[t] = *reinterpret_cast<const f32x4 *>(mine + r * 256 + ((er ^ r) * 16));
            }
            const unsigned c_off = (unsigned)(wave_u * 32 + i * 16 + eq) * ldc_b + cq * 256 + er * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) stb(c_rs, c_off, 4 * t * ldc_b, v[t]);
        }
        if (SPLIT_ABL & 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
    }
}

// planes[(ks * 3 + p) * R + 16 (r / 16)] + chunk wpos(r % 16, t / 8), element t % 8  =  piece p of S[r][ks * 32 + t]
// (S: R x Cc fp32, row stride ld; R % 16 == 0, Cc % 32 == 0)
__global__ __launch_bounds__(256) void pack_split_kernel(const float *__restrict__ S, int64_t ld, int R, int Cc, unsigned short *__restrict__ planes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)R * Cc) return;
    const int r = (int)(i / Cc), c = (int)(i % Cc);
    const float x = S[(int64_t)r * ld + c];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    const int ks = c >> 5, t = c & 31;
    const int64_t base = ((int64_t)ks * 3 * R + (r & ~15)) * 32 + wpos(r & 15, t >> 3) * 8 + (t & 7);
    planes[base] = __builtin_bit_cast(unsigned short, h);
    planes[base + (int64_t)R * 32] = __builtin_bit_cast(unsigned short, m);
    planes[base + (int64_t)2 * R * 32] = __builtin_bit_cast(unsigned short, l);
}
}  // namespace

static int g_split_persistent = 1;  // rlppo_dbg_set(36, 0/1): the split-bf16 products walked by persistent workgroups (large launches)
void set_split_persistent(int v) { g_split_persistent = v; }
bool nt_split_ok(int N, int K) { return N > 0 && N % 256 == 0 && K >= 32 && K % 32 == 0; }

int launch_pack_split(hipStream_t st, const float *S, int64_t ld, int R, int Cc, unsigned short *planes) {
    RLPPO_CHECK_ARG(S && planes && R > 0 && R % 16 == 0 && Cc > 0 && Cc % 32 == 0 && ld >= Cc, "pack_split: R=%d C=%d ld=%ld", R, Cc, (long)ld);
    const int64_t n = (int64_t)R * Cc;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, S, ld, R, Cc, planes);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// mode 0: C[M][N] = relu(A[M][K] . W^T + bias), bits <- [C > 0]; mode 1: C = (A . B^T) masked by bits (bias unused)
int launch_gemm_nt_split(hipStream_t st, const float *A, int64_t lda, const unsigned short *planes, const float *bias, float *C, int64_t ldc,
                         int64_t M, int N, int K, int mode, unsigned long long *bits) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(nt_split_ok(N, K) && A && planes && C && bits && (bias || mode == SPLIT_DX) && (mode == SPLIT_FWD || mode == SPLIT_DX) &&
                        lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0,
                    "gemm_nt_split: N=%d K=%d mode=%d lda=%ld ldc=%ld", N, K, mode, (long)lda, (long)ldc);
    const int64_t lim = (int64_t)1 << 31;
    RLPPO_CHECK_ARG(257 * lda * 4 < lim && 257 * ldc * 4 < lim && (int64_t)(K / 32) * 3 * N * 64 < lim, "gemm_nt_split: operand too wide for 32-bit offsets");
    static PerDeviceOnce attr[4];
    constexpr int LDS_BYTES = 2 * STAGE;
    const int rt128 = (int)cdiv(M, 128);
    const int row_tiles = (int)cdiv(M, 256), col_tiles = N / 256;
    int cus = 0;
    if (int rc_ = device_cu_count(&cus)) return rc_;
    if (g_split_persistent && (K / 32) % 2 == 0 && row_tiles >= 2 * (cus / (8 * col_tiles) > 0 ? cus / (8 * col_tiles) : 1) * 8 && col_tiles <= 8) {
        // persistent workgroups: whole groups of 8 row tiles x all column tiles, never more workgroups than CUs (or than the work)
        const int group = 8 * col_tiles;
        int grid = cus / group * group;
        const int need = (int)cdiv(row_tiles, 8) * group;
        grid = grid < group ? group : (grid > need ? need : grid);
        if (mode == SPLIT_FWD) {
            if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_split_p_kernel<SPLIT_FWD>, LDS_BYTES, attr[2])) return rc_;
            hipLaunchKernelGGL((gemm_nt_split_p_kernel<SPLIT_FWD>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st, A, (unsigned)(lda * 4), planes, N, bias,
                               C, (unsigned)(ldc * 4), M, K, bits, row_tiles, col_tiles, rt128);
        } else {
            if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_split_p_kernel<SPLIT_DX>, LDS_BYTES, attr[3])) return rc_;
            hipLaunchKernelGGL((gemm_nt_split_p_kernel<SPLIT_DX>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st, A, (unsigned)(lda * 4), planes, N, bias,
                               C, (unsigned)(ldc * 4), M, K, bits, row_tiles, col_tiles, rt128);
        }
        RLPPO_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)row_tiles, (unsigned)col_tiles);
    if (mode == SPLIT_FWD) {
        if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_split_kernel<SPLIT_FWD>, LDS_BYTES, attr[0])) return rc_;
        hipLaunchKernelGGL((gemm_nt_split_kernel<SPLIT_FWD>), grid, dim3(512), LDS_BYTES, st, A, (unsigned)(lda * 4), planes, N, bias, C,
                           (unsigned)(ldc * 4), M, K, bits, rt128);
    } else {
        if (int rc_ = set_dynamic_lds_once((const void *)gemm_nt_split_kernel<SPLIT_DX>, LDS_BYTES, attr[1])) return rc_;
        hipLaunchKernelGGL((gemm_nt_split_kernel<SPLIT_DX>), grid, dim3(512), LDS_BYTES, st, A, (unsigned)(lda * 4), planes, N, bias, C,
                           (unsigned)(ldc * 4), M, K, bits, rt128);
    }
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo
