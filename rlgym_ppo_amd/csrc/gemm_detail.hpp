// gemm_detail.hpp -- device helpers shared by the GEMM translation units (gemm.hip: fp32 MFMA kernels; gemm_b16.hip: the
// bf16-in-memory kernels of the bf16 update precision): buffer descriptors and 16-byte buffer accesses, the epilogue of the nt
// kernels, the ReLU bitmask, the XCD-aware tile order, the LDS swizzle.  Internal to csrc/.
#pragma once
#include "common.hpp"

namespace rlppo {


typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int SBM = 128;  // rows of a gemm_nt output tile
constexpr int EPI_NONE = 99;  // nt_epilogue: store the accumulators as they are
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 128-bit buffer descriptor over [base, base + bytes): base and bytes must be wave-uniform.  Loads past `bytes` return 0
// and stores past it are dropped by the hardware range check (used for the ragged last row tile).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
// buffer_load_dwordx4 v, voff, srsrc, soff offen: per-lane 32-bit offset + scalar offset, no vector address arithmetic
__device__ __forceinline__ f32x4 ldb(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void stb(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
// relu in ONE compiler-visible instruction: v_med3_f32(x, 0, +inf).  (x > 0 ? x : 0 compiles to two v_max because of NaN
// canonicalisation; an inline-asm v_max is invisible to the MFMA -> VALU hazard recogniser and read accumulators early:
// rare wrong activations under load, found by the 2-rank test.)  NaN -> 0 like the select form.
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
}  // namespace

// The 128-column tile of a wave (32 rows) leaves through LDS, 16 rows at a time: a lane holds 4 consecutive columns of one row (a
// wave-instruction would store 16 rows x 64 bytes); parked in 8 KiB of the wave's own (16-byte chunk c of row r at chunk c ^ r) and
// read back as rows it leaves as 2 rows x 512 contiguous bytes per wave-instruction.  In-wave: no barrier.  `park` = this wave's
// 8 KiB of the workgroup's stage buffers, free once the K loop's last barrier has passed.  [r4] Against the direct form at
// M = 524,288: first layer 0.288 -> 0.272 ms, dX head 0.253 -> 0.231, dX hidden 0.514 -> 0.503, forward hidden 0.515 -> 0.509
// (the launches whose output is most of their traffic gain most): 7.81 -> 7.69 ms of GEMM launches per pass.
__device__ __forceinline__ void nt_store_parked(f32x4 (&acc)[2][8], char *park, float *C, unsigned ldc_b, int64_t m0, int n0, int rows_here,
                                                int wave, int lane) {
#ifndef NT_ABL  // -DNT_ABL=16: timing-only build whose 128-column tile stores are dropped by the descriptor (profiles/r04_nt_parked_stores.txt)
#define NT_ABL 0
#endif
    const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)n0 * 4, (NT_ABL & 16) ? 0u : (unsigned)(rows_here - 1) * ldc_b + 128 * 4);
    const int r16 = lane & 15, q = lane >> 4, rr = lane >> 5, c32 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the first half's read-back is in registers
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(park + r16 * 512 + (((j * 4 + q) ^ r16) * 16)) = acc[i][j];
        f32x4 v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = 2 * t + rr;
            v[t] = *reinterpret_cast<const f32x4 *>(park + r * 512 + ((c32 ^ r) * 16));
        }
        const unsigned c_off = (unsigned)(wave * 32 + i * 16 + rr) * ldc_b + c32 * 16;
#pragma unroll
        for (int t = 0; t < 8; ++t) stb(c_rs, c_off, 2 * t * ldc_b, v[t]);
    }
}

// epilogue shared by the nt kernels: lane owns C[m0 + wave*32 + 16 i + r16][n0 + 16 j + 4 q + (0..3)]; rows past M fall
// outside the descriptor and are dropped by the range check
template <int NB, int EPI>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[2][NB], const float *mask_src, unsigned ldm_b, float *C, unsigned ldc_b,
                                            int64_t m0, int n0, int rows_here, int wave, int r16, int q, char *park = nullptr) {
    constexpr int BN = NB * 16;
    const int row_l = wave * 32 + r16;
    const __amdgpu_buffer_rsrc_t c_rs = make_rsrc(reinterpret_cast<char *>(C) + m0 * ldc_b + (int64_t)n0 * 4,
                                                  (unsigned)(rows_here - 1) * ldc_b + BN * 4);
    const unsigned c_off = (unsigned)row_l * ldc_b + q * 16;
    // The activation is applied IN PLACE over the accumulators first and the stores are issued afterwards, from registers
    // that nothing writes again.  With a shared temporary (store v[0:3]; next v_max overwrites v0..v3) the 16-byte buffer
    // stores were seen to pick up the NEXT block's values when the memory pipeline is backed up by another process
    // (scratch/stress_nt.py; the compiler's hazard table treats a buffer store with an SGPR soffset as safe to overwrite).
    if (EPI == EPI_MASK) {
        const __amdgpu_buffer_rsrc_t m_rs = make_rsrc(reinterpret_cast<const char *>(mask_src) + m0 * ldm_b + (int64_t)n0 * 4,
                                                      (unsigned)(rows_here - 1) * ldm_b + BN * 4);
        const unsigned m_off = (unsigned)row_l * ldm_b + q * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const f32x4 h = ldb(m_rs, m_off, 16 * i * ldm_b + j * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = h[e] > 0.f ? acc[i][j][e] : 0.f;
            }
    } else if (EPI != EPI_BIAS && EPI != EPI_NONE) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (EPI == EPI_BIAS_RELU) acc[i][j][e] = relu1(acc[i][j][e]);
                    if (EPI == EPI_BIAS_TANH) acc[i][j][e] = tanhf(acc[i][j][e]);
                }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NB == 8) {
        if (park) {  // (wave-uniform) 128-column tiles leave as 2 rows x 512 contiguous bytes per wave-instruction
            nt_store_parked(acc, park, C, ldc_b, m0, n0, rows_here, wave, q * 16 + r16);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) stb(c_rs, c_off, 16 * i * ldc_b + j * 64, acc[i][j]);
}

// ReLU bitmask (rlppo_dbg_set(19)).  dX = (dY . W) masked by [h > 0] needs one BIT of the forward activation per element,
// but re-reading h costs a 64 KB tile per workgroup whose latency nothing hides (the accumulators occupy the registers the
// tile would have to be prefetched into): the masked epilogue is 11 % of a dX launch (scratch/epi_cost.py).  Instead the
// forward epilogue of a hidden layer also emits, per lane, the 64 bits [acc > 0] of the 64 outputs the lane owns --
// bit (i * NB + j) * 4 + e for C[.. + 16 i + r16][.. + 16 j + 4 q + e] -- as one 8-byte word at
// bits[(row_tile * col_tiles + col_tile) * 256 + tid]; the dX kernel of the same tile geometry loads its word before the
// K loop (2 VGPRs) and the epilogue is 2 VALU instructions per element with no memory access.  1/32 of the bytes of h.
template <int NB>
__device__ __forceinline__ unsigned long long relu_bits(const f32x4 (&acc)[2][NB]) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int b = (i * NB + j) * 4 + e;
                const unsigned v = acc[i][j][e] > 0.f ? 1u : 0u;
                if (b < 32) lo |= v << b;
                else hi |= v << (b - 32);
            }
    return ((unsigned long long)hi << 32) | lo;
}
template <int NB>
__device__ __forceinline__ void apply_bits(f32x4 (&acc)[2][NB], unsigned long long w) {
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int b = (i * NB + j) * 4 + e;
                const unsigned m = 0u - (((b < 32 ? lo : hi) >> (b & 31)) & 1u);  // 0 or ~0
                const float x = acc[i][j][e];  // a scalar copy: bit-casting the vector ELEMENT expression reads element 0
                acc[i][j][e] = __uint_as_float(__float_as_uint(x) & m);
            }
}

// XCD-aware tile order.  Workgroups are dispatched in linear id order (x fastest) and id i runs on XCD i % 8, each XCD with
// its own L2.  The column tiles of one row tile read the same A rows, so they should run on the same XCD at about the same
// time: ids are taken in groups of 8 * (column tiles); within a group, id g -> row tile 8*group + g % 8, column tile g / 8.
// With row tiles as the fast index instead, the second reader of an A tile came 512-4096 workgroups later and, once A no
// longer fitted the 256 MB memory-side cache (fused minibatches), from HBM again.
// [r6] A row count that is not a multiple of 8 tiles (the reference's own batch: 50,000 rows = 391 tiles) used to switch the order
// OFF altogether -- every A tile was then fetched by two XCDs: 1.52 x the operand bytes per hidden launch by PMC
// (profiles/r06_traffic_rows.txt).  Now the first 8 * (nr / 8) row tiles are ordered as above and only the last nr % 8 take the plain
// order (ids behind all grouped ones: row tile fastest).
__device__ __forceinline__ void xcd_tile(int &row_tile, int &col_tile) {
    const int nr = gridDim.x, nc = gridDim.y;
    row_tile = blockIdx.x;
    col_tile = blockIdx.y;
    if (nc > 1 && nr >= 8) {
        const int id = blockIdx.y * nr + blockIdx.x, full = nr & ~7, grouped = full * nc;
        if (id < grouped) {
            const int g = id % (8 * nc);
            row_tile = (id / (8 * nc)) * 8 + (g & 7);
            col_tile = g >> 3;
        } else {
            const int t = id - grouped, r = nr - full;
            row_tile = full + t % r;
            col_tile = t / r;
        }
    }
}

// ------------------------------------------------------------------------------------------------ gemm_nt, LDS-DMA staging
// Same tile, but the A/B tiles go global -> LDS directly (buffer_load_dwordx4 ... lds, 16 B per lane, 1 KiB per wave
// instruction): no staging registers, no ds_write pass, no vector instruction at all between the MFMA streams.  One wave
// instruction fills 64/CPR consecutive LDS rows; the XOR swizzle of the LDS image is applied to the SOURCE address (the LDS
// side of an LDS-DMA is lane-linear).  BKT = 16 halves the LDS footprint (32 KiB) so that 4 workgroups share a CU.
template <int BKT>
__device__ __forceinline__ int dswz(int row, int chunk) {
    if (BKT == 32) return row * 32 + ((chunk ^ (row & 7)) << 2);
    return row * 16 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 2);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// host side, shared by the two weight-gradient launchers (defined in gemm.hip / gemm_b16.hip)
int64_t tn_partial_rows(int out, int in, int64_t M);           // rows per workgroup of the 128 x 128-tile forms
int64_t tn_partial_rows_wide(int pout, int pin, int64_t M);    // ... of the 256 x 256-tile bf16 form
struct TnAlt {     // device-side second operand set of a paired gemm_tn launch
    const float *dY, *X;
    float *partial;
    int interleave;  // 1: ids of both products' tiles interleave within a row split (same XCD back to back: a gathered first layer's two
                     // products read the same rows of the experience buffer) instead of the second product taking the upper half of z
};
struct TnRedAlt {  // ... and of its reduction
    const float *partial;
    float *dW, *db;
};
int launch_tn_reduce(hipStream_t st, const float *partial, int splits, int tiles_x, int tiles_y, float *dW, float *db, int out, int in,
                     int ni = 4, int nj = 4, int wn = 2, const TnRedAlt *alt = nullptr);  // tile geometry of the producing kernel (default: 128 x 128)
}  // namespace rlppo
