// gemm.hip -- the fp32 MFMA GEMM kernels of the MLP forward / backward (gfx950, wave64), LDS-DMA staged, scalar-addressed.
//
// Why fp32 MFMA: BASELINE.json asks for fp32 losses/grads within 1e-5 relative of the reference's CPU path.
// v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain (MI355X_MICROARCH.md "Matrix cores") at the fp32 vector peak
// (157 TFLOP/s), so parity needs no error analysis and the VALU stays free for epilogues.
//
//   gemm_nt : C[M][N] = epi(A[M][K] . B[N][K]^T)  -- both operands contraction-contiguous.
//             forward:   A = activations, B = packed W[out][in]     (epi = bias / bias+relu / bias+tanh)
//             backward:  A = dY,          B = packed W^T[in][out]   (epi = relu mask: bitmask of the forward, or the saved activation)
//   gemm_tn : dW[N][K] += dY[M][N]^T . X[M][K], db[N] += colsum(dY) -- contraction over the row (sample) axis, split over
//             workgroups along M into partial 128 x 128 tiles that a reduction kernel sums in a fixed order into the flat
//             gradient arena (the reference accumulates minibatch gradients into .grad, ppo_learner.py:179-180).
//
// MFMA operand trick: a lane loads 4 consecutive k with ONE ds_read_b128 and feeds them to 4 successive 16x16x4 MFMAs;
// hardware k-slot (lane>>4) then covers k = 4*(lane>>4)+s in step s -- a permutation of the contraction order applied
// identically to both operands, so the product is unchanged.
//
// Scalar addressing (measured): while one wave of a SIMD streams MFMAs, a VALU instruction of ANOTHER wave
// takes ~400 cycles to issue (a buffer load or an LDS-DMA does not).  Every address here is therefore an SGPR buffer
// descriptor (tile origin + k advance, scalar ALU) + a 32-bit per-lane offset computed once; the A/B tiles go global -> LDS
// by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write pass); the bias is the accumulators' initial value.
// Operands must span < 2 GiB from a tile's first to its last addressed byte (host-checked).
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

// epilogue shared by the nt kernels: lane owns C[m0 + wave*32 + 16 i + r16][n0 + 16 j + 4 q + (0..3)]; rows past M fall
// outside the descriptor and are dropped by the range check
template <int NB, int EPI>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[2][NB], const float *mask_src, unsigned ldm_b, float *C, unsigned ldc_b,
                                            int64_t m0, int n0, int rows_here, int wave, int r16, int q) {
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
__device__ __forceinline__ void xcd_tile(int &row_tile, int &col_tile) {
    const int nr = gridDim.x, nc = gridDim.y;
    row_tile = blockIdx.x;
    col_tile = blockIdx.y;
    if ((nr & 7) == 0 && nc > 1) {
        const int id = blockIdx.y * nr + blockIdx.x, g = id % (8 * nc);
        row_tile = (id / (8 * nc)) * 8 + (g & 7);
        col_tile = g >> 3;
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

template <int NB, int EPI, int BKT, bool BITS = false>
__global__ __launch_bounds__(256, BKT == 16 ? 4 : 2) void gemm_nt_dma_kernel(const float *__restrict__ A, unsigned lda_b,
                                                                              const float *__restrict__ B, unsigned ldb_b,
                                                                              const float *__restrict__ bias,
                                                                              const float *__restrict__ mask_src,
                                                                              unsigned ldm_b, float *__restrict__ C,
                                                                              unsigned ldc_b, int64_t M, int K,
                                                                              unsigned long long *__restrict__ bits = nullptr) {
    constexpr int BN = NB * 16;
    constexpr int CPR = BKT / 4;     // 16-byte chunks per tile row
    constexpr int RPW = 64 / CPR;    // tile rows one wave instruction fills
    constexpr int RPP = 4 * RPW;     // rows per pass of the 4 waves
    constexpr int A_IT = SBM / RPP, B_IT = BN / RPP;
    // a column tile that is not a whole number of staging passes (BN = 96 at BKT = 16: 1.5 passes of 64 rows) gets one more
    // pass issued only by the waves whose rows exist (wave-uniform branch): the 96-wide policy head then runs the BK = 16 /
    // four-workgroups-per-CU form like the 128-multiples instead of BK = 32 at two per CU
    constexpr int B_REM = BN % RPP;
    static_assert(B_REM % RPW == 0, "the ragged staging pass must be whole wave instructions");
    __shared__ __attribute__((aligned(16))) float lds[2 * SBM * BKT + 2 * BN * BKT];
    float *As = lds;
    float *Bs = lds + 2 * SBM * BKT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // provably uniform: it addresses the DMA destination (M0)
    const int r16 = lane & 15, q = lane >> 4;
    int row_tile, col_tile;
    xcd_tile(row_tile, col_tile);
    const int64_t m0 = (int64_t)row_tile * SBM;
    const int n0 = col_tile * BN;
    const int rows_here = (int)((M - m0) < SBM ? (M - m0) : SBM);
    unsigned long long *const bit_word = BITS ? bits + ((size_t)row_tile * gridDim.y + col_tile) * 256 + tid : nullptr;
    unsigned long long mask_word = 0;
    if (BITS && EPI == EPI_MASK) mask_word = *bit_word;  // requested before the K loop: long arrived when the epilogue needs it

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b,
                                                  (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(reinterpret_cast<const char *>(B) + (int64_t)n0 * ldb_b,
                                                  (unsigned)(BN - 1) * ldb_b + (unsigned)K * 4);
    // lane -> (row, physical chunk) of the 1 KiB piece its wave instruction fills; it fetches the logical chunk that the
    // swizzle maps there (rows of later passes keep the swizzle key, so one offset serves all passes)
    const int row_p = wave * RPW + lane / CPR, pch = lane % CPR;
    const int lch = BKT == 32 ? (pch ^ (row_p & 7)) : (pch ^ ((0 - (row_p >> 2)) & 3));
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    const unsigned b_off = (unsigned)row_p * ldb_b + lch * 16;
    const unsigned a_step = (unsigned)RPP * lda_b, b_step = (unsigned)RPP * ldb_b;

    f32x4 acc[2][NB];
    if (EPI == EPI_MASK) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
        if (B_REM != 0 && wave_u * RPW < B_REM)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, Bd + B_IT * RPP * BKT, 16, b_off, kb + B_IT * b_step, 0, 0);
    };

    const int nk = K / BKT;
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if ((kt + 1) < nk) issue_tile(cur ^ 1, (unsigned)(kt + 1) * (BKT * 4));
        const float *Ac = As + cur * SBM * BKT + (wave * 32) * BKT;
        const float *Bc = Bs + cur * BN * BKT;
#pragma unroll
        for (int kc = 0; kc < BKT / 16; ++kc) {
            f32x4 fa[2], fb[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&Ac[dswz<BKT>(i * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&Bc[dswz<BKT>(j * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = MFMA16(fb[j][s], fa[i][s], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above the wait: they are what hides the DMA latency
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next tile have landed in LDS
        __syncthreads();
    }
    if (BITS && EPI == EPI_MASK) {
        apply_bits<NB>(acc, mask_word);
        nt_epilogue<NB, EPI_NONE>(acc, nullptr, 0, C, ldc_b, m0, n0, rows_here, wave, r16, q);
    } else if (BITS && EPI == EPI_BIAS_RELU) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = relu1(acc[i][j][e]);
        *bit_word = relu_bits<NB>(acc);
        nt_epilogue<NB, EPI_NONE>(acc, nullptr, 0, C, ldc_b, m0, n0, rows_here, wave, r16, q);
    } else {
        nt_epilogue<NB, EPI>(acc, mask_src, ldm_b, C, ldc_b, m0, n0, rows_here, wave, r16, q);
    }
}

// ------------------------------------------------------------------------------------------------ gemm_nt, bf16 operands
// Inference-only forward (BASELINE configs[4]: "bf16 fwd / fp32 master weights"): the fp32 activations and the fp32 master
// weights are staged exactly as in gemm_nt_dma_kernel (fp32 tiles in LDS), rounded to bf16 when a lane builds its MFMA
// operands, multiplied by v_mfma_f32_16x16x32_bf16 and accumulated / biased / activated in fp32.  Selected by
// rlppo_set_inference_precision(1) for the rollout entry points only; rlppo_ppo_minibatch never uses it.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 to_bf16x8(const float *lo, const float *hi) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(lo), b = *reinterpret_cast<const f32x4 *>(hi);
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = (__bf16)a[e];
        r[4 + e] = (__bf16)b[e];
    }
    return r;
}

template <int NB, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(const float *__restrict__ A, unsigned lda_b,
                                                                              const float *__restrict__ B, unsigned ldb_b,
                                                                              const float *__restrict__ bias,
                                                                              const float *__restrict__ mask_src,
                                                                              unsigned ldm_b, float *__restrict__ C,
                                                                              unsigned ldc_b, int64_t M, int K) {
    constexpr int BN = NB * 16;
    constexpr int BKT = 32;  // one v_mfma_f32_16x16x32_bf16 step per LDS tile
    constexpr int CPR = BKT / 4;     // 16-byte chunks per tile row
    constexpr int RPW = 64 / CPR;    // tile rows one wave instruction fills
    constexpr int RPP = 4 * RPW;     // rows per pass of the 4 waves
    constexpr int A_IT = SBM / RPP, B_IT = BN / RPP;
    static_assert(BN % RPP == 0, "column tile must be a whole number of staging passes");
    __shared__ __attribute__((aligned(16))) float lds[2 * SBM * BKT + 2 * BN * BKT];
    float *As = lds;
    float *Bs = lds + 2 * SBM * BKT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // provably uniform: it addresses the DMA destination (M0)
    const int r16 = lane & 15, q = lane >> 4;
    int row_tile, col_tile;
    xcd_tile(row_tile, col_tile);
    const int64_t m0 = (int64_t)row_tile * SBM;
    const int n0 = col_tile * BN;
    const int rows_here = (int)((M - m0) < SBM ? (M - m0) : SBM);

    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b,
                                                  (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(reinterpret_cast<const char *>(B) + (int64_t)n0 * ldb_b,
                                                  (unsigned)(BN - 1) * ldb_b + (unsigned)K * 4);
    // lane -> (row, physical chunk) of the 1 KiB piece its wave instruction fills; it fetches the logical chunk that the
    // swizzle maps there (rows of later passes keep the swizzle key, so one offset serves all passes)
    const int row_p = wave * RPW + lane / CPR, pch = lane % CPR;
    const int lch = BKT == 32 ? (pch ^ (row_p & 7)) : (pch ^ ((0 - (row_p >> 2)) & 3));
    const unsigned a_off = (unsigned)row_p * lda_b + lch * 16;
    const unsigned b_off = (unsigned)row_p * ldb_b + lch * 16;
    const unsigned a_step = (unsigned)RPP * lda_b, b_step = (unsigned)RPP * ldb_b;

    f32x4 acc[2][NB];
    if (EPI == EPI_MASK) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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

    const int nk = K / BKT;
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if ((kt + 1) < nk) issue_tile(cur ^ 1, (unsigned)(kt + 1) * (BKT * 4));
        const float *Ac = As + cur * SBM * BKT + (wave * 32) * BKT;
        const float *Bc = Bs + cur * BN * BKT;
        // lane (r16, q) holds k = 8 q .. 8 q + 7 of its row: chunks 2 q and 2 q + 1 of the swizzled fp32 image, rounded to
        // bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) as they become MFMA operands
        bf16x8 fa[2], fb[NB];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = to_bf16x8(&Ac[dswz<32>(i * 16 + r16, 2 * q)], &Ac[dswz<32>(i * 16 + r16, 2 * q + 1)]);
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[j] = to_bf16x8(&Bc[dswz<32>(j * 16 + r16, 2 * q)], &Bc[dswz<32>(j * 16 + r16, 2 * q + 1)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above the wait: they are what hides the DMA latency
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next tile have landed in LDS
        __syncthreads();
    }
    nt_epilogue<NB, EPI>(acc, mask_src, ldm_b, C, ldc_b, m0, n0, rows_here, wave, r16, q);
}

template <int NB>
static int launch_nt_1(hipStream_t st, dim3 grid, int epi, const float *A, unsigned lda_b, const float *B, unsigned ldb_b,
                       const float *bias, const float *mask_src, unsigned ldm_b, float *C, unsigned ldc_b, int64_t M,
                       int K) {
    // BK = 16 (<= 32 KiB of LDS, four workgroups per CU) fills 64 tile rows per staging pass (+ a ragged pass of 32 for the
    // 96-wide tile); the 32-wide tile (half a pass) stays at BK = 32
    constexpr int BKT = (NB * 16) % 32 == 0 && NB * 16 >= 64 ? 16 : 32;
#define NT(E)                                                                                                          \
    case E:                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_dma_kernel<NB, E, BKT>), grid, dim3(256), 0, st, A, lda_b, B, ldb_b, bias, mask_src, \
                           ldm_b, C, ldc_b, M, K);                                                                     \
        break;
    switch (epi) {
        NT(EPI_BIAS) NT(EPI_BIAS_RELU) NT(EPI_BIAS_TANH) NT(EPI_MASK)
        default:
            set_error("gemm_nt: bad epilogue %d", epi);
            return RLPPO_ERR_ARG;
    }
#undef NT
    RLPPO_LAUNCH_CHECK();
    return 0;
}

static int g_infer_bf16 = 0;  // rlppo_set_inference_precision
void set_infer_bf16(int v) { g_infer_bf16 = v; }
int get_infer_bf16() { return g_infer_bf16; }

template <int NB>
static int launch_bf16_1(hipStream_t st, dim3 grid, int epi, const float *A, unsigned lda_b, const float *B, unsigned ldb_b,
                         const float *bias, float *C, unsigned ldc_b, int64_t M, int K) {
#define BF(E)                                                                                                          \
    case E:                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<NB, E>), grid, dim3(256), 0, st, A, lda_b, B, ldb_b, bias, nullptr, 0u, C, \
                           ldc_b, M, K);                                                                               \
        break;
    switch (epi) {
        BF(EPI_BIAS) BF(EPI_BIAS_RELU) BF(EPI_BIAS_TANH)
        default:
            set_error("gemm_nt (bf16 operands): epilogue %d is not a forward epilogue", epi);
            return RLPPO_ERR_ARG;
    }
#undef BF
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// forward product with bf16-rounded operands; same argument checks as launch_gemm_nt (done by the caller)
int launch_gemm_nt_bf16(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int N, int nb, int K, int epi) {
    dim3 grid((unsigned)cdiv(M, SBM), (unsigned)(N / (nb * 16)));
    const unsigned la = (unsigned)(lda * 4), lb = (unsigned)(ldb * 4), lc = (unsigned)(ldc * 4);
    switch (nb) {
        case 8: return launch_bf16_1<8>(st, grid, epi, A, la, B, lb, bias, C, lc, M, K);
        case 6: return launch_bf16_1<6>(st, grid, epi, A, la, B, lb, bias, C, lc, M, K);
        case 4: return launch_bf16_1<4>(st, grid, epi, A, la, B, lb, bias, C, lc, M, K);
        default: return launch_bf16_1<2>(st, grid, epi, A, la, B, lb, bias, C, lc, M, K);
    }
}

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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
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

// Applicability of the bf16-in-memory forward: K a multiple of 64 (one LDS tile row = 64 k-values), hidden widths a multiple
// of 128 (the bitmask tile geometry).  Everything else takes the fp32 kernels on the ROUNDED fp32 copies -- the same products
// (a product of two bf16 values is exact in fp32), only slower -- followed by round_rows (optim.hip).
static int g_b16_wide = 1;  // rlppo_dbg_set(23, .): 256 x 256 tiles for the hidden / dX products of the bf16 update precision (0: 128 x 128)
void set_b16_wide_tiles(int on) { g_b16_wide = on != 0; }
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
        static bool attr_set[2] = {false, false};
        const int which = mode == B16_DX ? 1 : 0;
        const int rt128 = (int)cdiv(M, 128);
#define B16W(MODE_)                                                                                                          \
    do {                                                                                                                     \
        constexpr int LDS_BYTES = 2 * (256 + 256) * 32 * 4; /* two 64 KiB stages; the epilogue parks 16 KiB per wave in them */ \
        if (!attr_set[which]) {                                                                                              \
            RLPPO_HIP(hipFuncSetAttribute((const void *)gemm_nt_b16w_kernel<MODE_, 256, 32, 2>,                               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));                           \
            attr_set[which] = true;                                                                                          \
        }                                                                                                                    \
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

// floats of workspace one hidden layer's ReLU bitmask needs (8 bytes per lane and 128 x 128 tile); 0 = width not supported
size_t nt_bits_floats(int64_t M, int N) {
    if (N % 128 != 0 || M <= 0) return 0;
    return (size_t)cdiv(M, SBM) * (size_t)(N / 128) * 256 * 2;
}
// The hidden-layer forward (epi = EPI_BIAS_RELU: writes `bits`) or the masked dX product (epi = EPI_MASK: reads `bits`
// instead of the activation) through gemm_nt_dma_kernel<8, ., 16, true>.  Returns -1 when that form does not apply
// (width not a multiple of 128, operands too wide for 32-bit tile offsets): the caller then uses launch_gemm_nt
// and, for the forward, must not hand the bitmask to the backward pass.
int launch_gemm_nt_bits(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int N, int K, int epi, unsigned long long *bits) {
    if (!bits || N % 128 != 0 || K % 16 != 0 || M <= 0) return -1;
    if (epi != EPI_BIAS_RELU && epi != EPI_MASK) return -1;
    const int64_t lim = (int64_t)1 << 31;
    if (129 * lda * 4 >= lim || 129 * ldb * 4 >= lim || 129 * ldc * 4 >= lim) return -1;
    dim3 grid((unsigned)cdiv(M, SBM), (unsigned)(N / 128));
    const unsigned la = (unsigned)(lda * 4), lb = (unsigned)(ldb * 4), lc = (unsigned)(ldc * 4);
    if (epi == EPI_BIAS_RELU)
        hipLaunchKernelGGL((gemm_nt_dma_kernel<8, EPI_BIAS_RELU, 16, true>), grid, dim3(256), 0, st, A, la, B, lb, bias, nullptr,
                           0u, C, lc, M, K, bits);
    else
        hipLaunchKernelGGL((gemm_nt_dma_kernel<8, EPI_MASK, 16, true>), grid, dim3(256), 0, st, A, la, B, lb, nullptr, nullptr, 0u,
                           C, lc, M, K, bits);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_nt(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                   const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int N, int K, int epi,
                   int bf16_operands) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(K > 0 && K % 32 == 0, "gemm_nt: K=%d must be a positive multiple of 32", K);
    RLPPO_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && lda >= K && ldb >= K && ldc >= N,
                    "gemm_nt: leading dimensions lda=%ld ldb=%ld ldc=%ld incompatible with K=%d N=%d", (long)lda,
                    (long)ldb, (long)ldc, K, N);
    RLPPO_CHECK_ARG(epi != EPI_MASK || (mask_src && ld_mask >= N && ld_mask % 4 == 0), "gemm_nt: mask operand missing");
    RLPPO_CHECK_ARG(epi == EPI_MASK || bias, "gemm_nt: bias operand missing");
    int nb;
    if (N % 128 == 0) nb = 8;
    else if (N == 96) nb = 6;
    else if (N == 64) nb = 4;
    else if (N == 32) nb = 2;
    else {
        set_error("gemm_nt: N=%d is not a padded output width", N);
        return RLPPO_ERR_ARG;
    }
    // offsets are relative to the workgroup's tile origin, so the 32-bit range only limits the leading dimensions
    const int64_t lim = (int64_t)1 << 31;
    RLPPO_CHECK_ARG(129 * lda * 4 < lim && 129 * ldb * 4 < lim && 129 * ldc * 4 < lim && 129 * ld_mask * 4 < lim,
                    "gemm_nt: a leading dimension is too wide for 32-bit tile offsets");
    if (bf16_operands && epi != EPI_MASK) return launch_gemm_nt_bf16(st, A, lda, B, ldb, bias, C, ldc, M, N, nb, K, epi);
    dim3 grid((unsigned)cdiv(M, SBM), (unsigned)(N / (nb * 16)));
    const unsigned la = (unsigned)(lda * 4), lb = (unsigned)(ldb * 4), lc = (unsigned)(ldc * 4), lm = (unsigned)(ld_mask * 4);
    switch (nb) {
        case 8: return launch_nt_1<8>(st, grid, epi, A, la, B, lb, bias, mask_src, lm, C, lc, M, K);
        case 6: return launch_nt_1<6>(st, grid, epi, A, la, B, lb, bias, mask_src, lm, C, lc, M, K);
        case 4: return launch_nt_1<4>(st, grid, epi, A, la, B, lb, bias, mask_src, lm, C, lc, M, K);
        default: return launch_nt_1<2>(st, grid, epi, A, la, B, lb, bias, mask_src, lm, C, lc, M, K);
    }
}

namespace {
constexpr int TM = 32;                 // sample rows per LDS stage
constexpr unsigned OOR = 0x80000000u;  // per-lane offset that fails every descriptor's range check
}  // namespace

// ------------------------------------------------------------------------------------------------ gemm_tn, LDS-DMA staging
// The dY / X stages go global -> LDS by buffer_load ... lds: a 1 KiB piece = 2 stage rows of 128 floats.  The LDS image is
// unpadded; rows m..m+3 of one 16-float column group (what a ds_read_b32 fragment read touches) are spread over the banks
// by XOR-ing the 16-byte chunk index with (m & 3) << 2, applied on the SOURCE address.  db column sums are read back from
// the staged dY tile (only by the workgroups of the first k tile).
//
// No atomics: 64 fp32 atomics per lane into dW cost 20-50 us of an 86 us launch (128-256 workgroups adding into the same
// 110-256 KB serialise in the memory-side atomic units).  The workgroup stores its 128 x 128 partial tile with 16 coalesced
// 16-byte stores per lane into partial[split][tile][(i*4+j)*256 + tid] and tn_reduce_kernel sums the splits afterwards: the
// sum order is fixed, so the weight gradients are bit-reproducible from run to run.
template <int TMT>
__global__ __launch_bounds__(256, 2) void gemm_tn_dma_kernel(const float *__restrict__ dY, unsigned ldy_b,
                                                                              int ny_valid, const float *__restrict__ X,
                                                                              unsigned ldx_b, int kx_valid,
                                                                              bool with_db, int out, int in, int64_t M,
                                                                              int rows_per_wg, float *__restrict__ partial) {
    constexpr int PPW = TMT / 8;  // 1 KiB pieces per wave, per operand and stage
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TMT * 128 < 8 * 128 ? 8 * 128 : 2 * 2 * TMT * 128];
    float *Ys = lds;                 // [2][TMT][128]
    float *Xs = lds + 2 * TMT * 128;  // [2][TMT][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    // XCD-aware order (see xcd_tile): the output tiles of one row split read the same dY / X rows, so they are given ids
    // that land on the same XCD back to back: id -> tile = (id % (8 T)) / 8, split = 8 (id / (8 T)) + id % 8
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
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
    const int ny_here = (ny_valid - n0) < 128 ? (ny_valid - n0) : 128;
    const int kx_here = (kx_valid - k0) < 128 ? (kx_valid - k0) : 128;

    const __amdgpu_buffer_rsrc_t y_rs = make_rsrc(reinterpret_cast<const char *>(dY) + mbeg * ldy_b + (int64_t)n0 * 4,
                                                  (unsigned)(rows - 1) * ldy_b + (unsigned)ny_here * 4);
    const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(reinterpret_cast<const char *>(X) + mbeg * ldx_b + (int64_t)k0 * 4,
                                                  (unsigned)(rows - 1) * ldx_b + (unsigned)kx_here * 4);
    // DMA lane map: piece = rows 2 (wave + 4 i) + (lane >> 5); physical chunk lane & 31 holds logical chunk ^ ((row & 3) << 2)
    const int row_l = 2 * wave + (lane >> 5);
    const int lch = (lane & 31) ^ ((row_l & 3) << 2);
    const unsigned y_off = (lch * 4 < ny_here) ? (unsigned)row_l * ldy_b + lch * 16 : OOR;
    const unsigned x_off = (lch * 4 < kx_here) ? (unsigned)row_l * ldx_b + lch * 16 : OOR;
    const unsigned y_row8 = 8u * ldy_b, x_row8 = 8u * ldx_b, y_stage = TMT * ldy_b, x_stage = TMT * ldx_b;  // uniform

    auto issue_stage = [&](int buf, int stage) {
        float *Yd = Ys + (buf * TMT + 2 * wave_u) * 128, *Xd = Xs + (buf * TMT + 2 * wave_u) * 128;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rs, Yd + 8 * i * 128, 16, y_off, (unsigned)stage * y_stage + i * y_row8, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, Xd + 8 * i * 128, 16, x_off, (unsigned)stage * x_stage + i * x_row8, 0, 0);
        }
    };
    // a ragged last stage: the DMA drops the rows past the split, so their (stale) LDS rows are cleared by hand
    const int zr = tid >> 5, zc = (tid & 31) * 4;
    auto clear_tail = [&](int buf) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = rem + zr; r < TMT; r += 8) {
            *reinterpret_cast<f32x4 *>(&Ys[(buf * TMT + r) * 128 + zc]) = z;
            *reinterpret_cast<f32x4 *>(&Xs[(buf * TMT + r) * 128 + zc]) = z;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bs4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_db = with_db && by == 0;

    issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (steps == 1 && rem < TMT) clear_tail(0);
    __syncthreads();
    // fragment addresses: element (m, col) lives at m*128 + (col ^ ((m & 3) << 4)); m & 3 == q for every fragment read
    int fy[4], fx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fy[i] = q * 128 + wn * 64 + ((i ^ q) << 4) + r16;
        fx[i] = q * 128 + wk * 64 + ((i ^ q) << 4) + r16;
    }
    const int db_off = zr * 128 + zc;  // db partial sums: rows zr + 8 i, physical chunk tid & 31
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        const bool more = (st + 1) < steps;
        if (more) issue_stage(cur ^ 1, st + 1);
        const float *Yc = Ys + cur * TMT * 128;
        const float *Xc = Xs + cur * TMT * 128;
        if (want_db) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) bs4 += *reinterpret_cast<const f32x4 *>(&Yc[db_off + 8 * i * 128]);
        }
#pragma unroll
        for (int c = 0; c < TMT / 16; ++c) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = c * 16 + s * 4;  // + q
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = Yc[m * 128 + fy[i]];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = Xc[m * 128 + fx[j]];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(fa[i], fb[j], acc[i][j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above the wait: they are what hides the DMA latency
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more && (st + 2) == steps && rem < TMT) clear_tail(cur ^ 1);
        __syncthreads();
    }

    if (want_db) {  // 8 threads (tid >> 5) hold partial sums of the same 4 columns: fold them through LDS
        float *red = lds;  // [8][128]; the staging buffers are dead (the loop ended with a barrier)
        const int lcol = ((tid & 31) ^ ((zr & 3) << 2)) * 4;  // the logical columns this thread's physical chunk holds
        *reinterpret_cast<f32x4 *>(&red[zr * 128 + lcol]) = bs4;
        __syncthreads();
        if (tid < 128 && (n0 + tid) < out) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) sum += red[r * 128 + tid];
            // column sums of this split: partial_db[split][n tile][128], behind the tile partials
            partial[(size_t)gridDim.z * gridDim.y * gridDim.x * (128 * 128) + ((size_t)bz * gridDim.x + bx) * 128 + tid] = sum;
        }
    }
    // The partial-tile stores are the LAST instructions of the wave: 16-byte buffer stores whose data registers are written
    // again soon afterwards can pick up the new values (the hazard of section 5 / tests/test_gpu_stress.py); here the
    // accumulators are never touched after them.
    {
        const size_t tile_id = (size_t)bz * (gridDim.x * gridDim.y) + (size_t)by * gridDim.x + bx;
        const __amdgpu_buffer_rsrc_t p_rs = make_rsrc(partial + tile_id * (128 * 128), 128 * 128 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) stb(p_rs, (unsigned)tid * 16, (unsigned)(i * 4 + j) * 4096, acc[i][j]);
    }
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

// Sums the partial tiles of gemm_tn_dma_kernel over the splits and adds the result into dW[out][in].
// Block = 64 consecutive 16-byte elements of one tile x 4 split lanes (one wave each: 1 KiB coalesced per load, 8 loads
// in flight); the four partial sums meet in LDS.  The final add is an atomic only so that launches of different
// minibatches that share dW stay safe; there is exactly one per element and launch.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float *__restrict__ partial, int splits, int tiles_x, int tiles,
                                                        float *__restrict__ dW, float *__restrict__ db, int out, int in) {
    __shared__ __attribute__((aligned(16))) float red[3][64][4];
    const int tile = blockIdx.y, e64 = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int elem4 = blockIdx.x * 64 + e64;  // 16-byte element of the tile: (i*4+j)*256 + tid
    const f32x4 *base = reinterpret_cast<const f32x4 *>(partial) + (size_t)tile * 4096 + elem4;
    const size_t stride = (size_t)tiles * 4096;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = sl;
    for (; sp + 28 < splits; sp += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + (size_t)(sp + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sp < splits; sp += 4) acc += __builtin_nontemporal_load(base + (size_t)sp * stride);
    if (sl > 0) *reinterpret_cast<f32x4 *>(&red[sl - 1][e64][0]) = acc;
    // bias gradient (one block per n tile): 128 columns x 2 halves of the splits, 16 loads in flight per thread, every
    // partial combined in a fixed order
    __shared__ float dbh[128];
    const bool db_block = db && blockIdx.x == 63 && tile < tiles_x;
    float sdb = 0.f;
    if (db_block) {
        const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
        const float *pdb = partial + (size_t)splits * tiles * (128 * 128) + (size_t)tile * 128 + c;
        const size_t dstride = (size_t)tiles_x * 128;
        const int s_lo = half ? (splits + 1) / 2 : 0, s_hi = half ? splits : (splits + 1) / 2;
        float a16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a16[u] = 0.f;
        int s2 = s_lo;
        for (; s2 + 15 < s_hi; s2 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = pdb[(size_t)(s2 + u) * dstride];
#pragma unroll
            for (int u = 0; u < 16; ++u) a16[u] += v[u];
        }
        for (int u = 0; s2 < s_hi; ++s2, ++u) a16[u] += pdb[(size_t)s2 * dstride];
#pragma unroll
        for (int u = 0; u < 16; ++u) sdb += a16[u];
        if (half) dbh[c] = sdb;
    }
    __syncthreads();
    if (db_block && threadIdx.x < 128 && tile * 128 + (int)threadIdx.x < out) atomicAdd(db + tile * 128 + threadIdx.x, sdb + dbh[threadIdx.x]);
    if (sl > 0) return;
#pragma unroll
    for (int r = 0; r < 3; ++r) acc += *reinterpret_cast<const f32x4 *>(&red[r][e64][0]);
    const int ij = elem4 >> 8, t = elem4 & 255;
    const int i = ij >> 2, j = ij & 3, wave = t >> 6, lane = t & 63;
    const int n0 = (tile % tiles_x) * 128, k0 = (tile / tiles_x) * 128;
    const int k = k0 + (wave & 1) * 64 + j * 16 + (lane & 15);
    const int nb = n0 + (wave >> 1) * 64 + i * 16 + (lane >> 4) * 4;
    if (k < in) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (nb + e < out) atomicAdd(dW + (size_t)(nb + e) * in + k, acc[e]);
    }
}

// rows per workgroup: no atomic traffic to trade against, so simply two workgroups per CU
static int64_t tn_partial_rows(int out, int in, int64_t M) {
    const int64_t tiles = cdiv(out, 128) * cdiv(in, 128);
    int64_t rows = tiles >= 2 ? 512 : 256;
    if (M < 64 * rows) rows = round_up(cdiv(M, 64) > 32 ? cdiv(M, 64) : 32, 32);
    // large M (fused minibatches): one round of workgroups (2 per CU) instead of more and more splits -- fewer partial tiles
    // to write and to reduce.  M = 524,288, us per launch at 512 / 1024 / 2048 / 4096 rows: hidden 567 / 527 / 505 / 494,
    // first layer 313 / 284 / 267 / 287, policy head 305 / 277 / 262 / 284
    const int64_t few = round_up(cdiv(M * tiles, 512), 32);
    return few > rows ? few : rows;
}
// rows per workgroup of the 256 x 256-tile bf16 form: one round of one workgroup per CU
static int64_t tn_partial_rows_wide(int pout, int pin, int64_t M) {
    const int64_t tiles = (int64_t)(pout / 256) * (pin / 256);
    const int64_t rows = round_up(cdiv(M * tiles, 256), 64);
    return rows > 256 ? rows : 256;
}
size_t tn_partial_floats(int out, int in, int64_t M) {
    if (M <= 0) return 0;
    size_t splits = (size_t)cdiv(M, tn_partial_rows(out, in, M));
    const int pout = (int)round_up(out, 128), pin = (int)round_up(in, 128);
    if (pout % 256 == 0 && pin % 256 == 0) {  // the wide bf16 form may split finer: size for whichever form splits more
        const size_t w = (size_t)cdiv(M, tn_partial_rows_wide(pout, pin, M));
        splits = w > splits ? w : splits;
    }
    return splits * (size_t)(cdiv(out, 128) * cdiv(in, 128)) * (128 * 128) + splits * (size_t)cdiv(out, 128) * 128;  // tiles + db
}

// dW[out][in] += dY^T . X, db[out] += colsum(dY) through partial tiles in `ws` (>= tn_partial_floats floats) + a reduction
int launch_gemm_tn(hipStream_t st, const float *dY, int64_t ldy, int ny_valid, const float *X, int64_t ldx, int kx_valid,
                   float *dW, float *db, int out, int in, int64_t M, float *ws, size_t ws_floats) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(ny_valid % 4 == 0 && kx_valid % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ny_valid <= ldy &&
                        kx_valid <= ldx && out <= ny_valid && in <= kx_valid,
                    "gemm_tn: bad shapes ny=%d kx=%d ldy=%ld ldx=%ld out=%d in=%d", ny_valid, kx_valid, (long)ldy,
                    (long)ldx, out, in);
    if (!ws || ws_floats < tn_partial_floats(out, in, M)) {
        set_error("gemm_tn: workspace %zu < %zu floats", ws ? ws_floats : (size_t)0, tn_partial_floats(out, in, M));
        return RLPPO_ERR_WORKSPACE;
    }
    const int rows_per_wg = (int)tn_partial_rows(out, in, M);
    const int64_t lim = (int64_t)1 << 30;
    RLPPO_CHECK_ARG((rows_per_wg + TM) * ldy * 4 < lim && (rows_per_wg + TM) * ldx * 4 < lim,
                    "gemm_tn: a leading dimension is too wide for 32-bit tile offsets");
    const int tiles_x = (int)cdiv(out, 128), tiles_y = (int)cdiv(in, 128), splits = (int)cdiv(M, rows_per_wg);
    dim3 grid((unsigned)tiles_x, (unsigned)tiles_y, (unsigned)splits);
    hipLaunchKernelGGL((gemm_tn_dma_kernel<TM>), grid, dim3(256), 0, st, dY, (unsigned)(ldy * 4), ny_valid, X,
                       (unsigned)(ldx * 4), kx_valid, db != nullptr, out, in, M, rows_per_wg, ws);
    RLPPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(64, (unsigned)(tiles_x * tiles_y)), dim3(256), 0, st, ws, splits, tiles_x,
                       tiles_x * tiles_y, dW, db, out, in);
    RLPPO_LAUNCH_CHECK();
    return 0;
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
        static bool attr_set = false;
        constexpr int LDS_BYTES = 2 * 4 * TNB_ROWS * 256;
        if (!attr_set) {
            RLPPO_HIP(hipFuncSetAttribute((const void *)gemm_tn_b16w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_tn_b16w_kernel, dim3((unsigned)(pout / 256), (unsigned)(pin / 256), (unsigned)splits), dim3(512), LDS_BYTES,
                           st, dY, (unsigned)(ldy * 2), X, (unsigned)(ldx * 2), db != nullptr, out, M, rows_w, ws);
    } else {
        dim3 grid((unsigned)tiles_x, (unsigned)tiles_y, (unsigned)splits);
        hipLaunchKernelGGL(gemm_tn_b16_kernel, grid, dim3(256), 0, st, dY, (unsigned)(ldy * 2), X, (unsigned)(ldx * 2), db != nullptr,
                           out, M, rows_per_wg, ws);
    }
    RLPPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(64, (unsigned)(tiles_x * tiles_y)), dim3(256), 0, st, ws, splits, tiles_x,
                       tiles_x * tiles_y, dW, db, out, in);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo
