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
#include "gemm_detail.hpp"

namespace rlppo {

// GATHER (the minibatch gather of experience_buffer.py:82-87 folded into the first layer's operand fetch): A is the
// experience buffer's state matrix and row r of the product is A[rowtab[r]] -- the DMA's descriptor is then a STRUCTURED buffer
// (stride = one buffer row) addressed by a per-lane row INDEX (buffer_load ... idxen offen lds): the index register is loaded once
// per tile from the pass's row table (u32 physical rows, written by gather_meta_kernel), the K advance stays the scalar offset,
// and the loop is instruction for instruction the contiguous one.  Rows past M read table entry "0" (the table's range check)
// -- a valid row whose products land in output rows the C descriptor drops.
template <int NB, int EPI, int BKT, bool BITS = false, bool GATHER = false>
__global__ __launch_bounds__(256, BKT == 16 ? 4 : 2) void gemm_nt_dma_kernel(const float *__restrict__ A, unsigned lda_b,
                                                                              const float *__restrict__ B, unsigned ldb_b,
                                                                              const float *__restrict__ bias,
                                                                              const float *__restrict__ mask_src,
                                                                              unsigned ldm_b, float *__restrict__ C,
                                                                              unsigned ldc_b, int64_t M, int K,
                                                                              unsigned long long *__restrict__ bits = nullptr,
                                                                              const unsigned *__restrict__ rowtab = nullptr,
                                                                              unsigned src_rows = 0, NtAlt alt = NtAlt{},
                                                                              NtDot dot = NtDot{}, NtDot dot_alt = NtDot{}) {
    // [r3] a launch may carry TWO products of the same shape (policy and critic layers of equal widths): blockIdx.z == 1 takes its
    // operands from `alt` (scalar selects; M, K, leading dimensions, row table are shared)
    // (the operand set is chosen below, once the tile is known)
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
    // Interleaved pairing: the grid's y range holds the column tiles of BOTH products, so the workgroups that read one row tile of
    // A -- for a gathered first layer both networks read the SAME rows of the experience buffer -- get consecutive ids on one XCD.
    int nct = gridDim.y;
    bool second = blockIdx.z != 0;
    if (alt.interleave) {
        nct >>= 1;
        second = col_tile >= nct;
        if (second) col_tile -= nct;
    }
    if (second) {  // (scalar selects)
        A = alt.A;
        B = alt.B;
        bias = alt.bias;
        C = alt.C;
        bits = alt.bits;
        dot = dot_alt;
    }
    const int64_t m0 = (int64_t)row_tile * SBM;
    const int n0 = col_tile * BN;
    const int rows_here = (int)((M - m0) < SBM ? (M - m0) : SBM);
    unsigned long long *const bit_word = BITS ? bits + ((size_t)row_tile * nct + col_tile) * 256 + tid : nullptr;
    unsigned long long mask_word = 0;
    if (BITS && EPI == EPI_MASK) mask_word = *bit_word;  // requested before the K loop: long arrived when the epilogue needs it

    const __amdgpu_buffer_rsrc_t a_rs = GATHER ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), (short)lda_b, src_rows, 0x00020000)
                                               : make_rsrc(reinterpret_cast<const char *>(A) + m0 * lda_b,
                                                           (unsigned)(rows_here - 1) * lda_b + (unsigned)K * 4);
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(reinterpret_cast<const char *>(B) + (int64_t)n0 * ldb_b,
                                                  (unsigned)(BN - 1) * ldb_b + (unsigned)K * 4);
    // lane -> (row, physical chunk) of the 1 KiB piece its wave instruction fills; it fetches the logical chunk that the
    // swizzle maps there (rows of later passes keep the swizzle key, so one offset serves all passes)
    const int row_p = wave * RPW + lane / CPR, pch = lane % CPR;
    const int lch = BKT == 32 ? (pch ^ (row_p & 7)) : (pch ^ ((0 - (row_p >> 2)) & 3));
    const unsigned a_off = GATHER ? (unsigned)lch * 16 : (unsigned)row_p * lda_b + lch * 16;
    const unsigned b_off = (unsigned)row_p * ldb_b + lch * 16;
    const unsigned a_step = (unsigned)RPP * lda_b, b_step = (unsigned)RPP * ldb_b;
    unsigned a_row[4];  // GATHER: the buffer rows of this lane's A_IT staging passes (a dependent bound loses the host-side launch stub: hipcc 7.2)
    static_assert(A_IT <= 4, "a_row");  // GATHER: the buffer row each of this lane's A_IT staging passes fetches
    if (GATHER) {
        const __amdgpu_buffer_rsrc_t t_rs = make_rsrc(rowtab + m0, (unsigned)rows_here * 4);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a_row[i] = __builtin_amdgcn_raw_buffer_load_b32(t_rs, (unsigned)(row_p + i * RPP) * 4, 0, 0);
    }

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
        for (int i = 0; i < A_IT; ++i) {
            if (GATHER)
                __builtin_amdgcn_struct_ptr_buffer_load_lds(a_rs, Ad + i * RPP * BKT, 16, a_row[i], a_off, kb, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, Ad + i * RPP * BKT, 16, a_off, kb + i * a_step, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, Bd + i * RPP * BKT, 16, b_off, kb + i * b_step, 0, 0);
        if (B_REM != 0 && wave_u * RPW < B_REM)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, Bd + B_IT * RPP * BKT, 16, b_off, kb + B_IT * b_step, 0, 0);
    };

    const int nk = K / BKT;
    if (GATHER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row indices are the first tile's addresses
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
    // the stage buffers are free (the loop's last barrier has passed): 8 KiB per wave for the tile's way out (nt_store_parked)
    char *const park = reinterpret_cast<char *>(lds) + wave_u * 8192;
    if (BITS && EPI == EPI_MASK) {
        apply_bits<NB>(acc, mask_word);
        nt_epilogue<NB, EPI_NONE>(acc, nullptr, 0, C, ldc_b, m0, n0, rows_here, wave, r16, q, park);
    } else if (BITS && EPI == EPI_BIAS_RELU) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = relu1(acc[i][j][e]);
        *bit_word = relu_bits<NB>(acc);
        if (dot.w) {  // (scalar condition) the one-output head that reads this layer: partial dot products of the tile's rows
            const __amdgpu_buffer_rsrc_t w_rs = make_rsrc(dot.w + n0, BN * 4);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const f32x4 wv = ldb(w_rs, (unsigned)(q * 16), j * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0 += acc[0][j][e] * wv[e];
                    s1 += acc[1][j][e] * wv[e];
                }
            }
            s0 += __shfl_xor(s0, 16);
            s1 += __shfl_xor(s1, 16);
            s0 += __shfl_xor(s0, 32);
            s1 += __shfl_xor(s1, 32);
            if (q == 0) {
                const float b0 = col_tile == 0 ? dot.b[0] : 0.f;
                const int r0 = wave * 32 + r16;
                if (r0 < rows_here) atomicAdd(dot.out + m0 + r0, s0 + b0);
                if (r0 + 16 < rows_here) atomicAdd(dot.out + m0 + r0 + 16, s1 + b0);
            }
        }
        nt_epilogue<NB, EPI_NONE>(acc, nullptr, 0, C, ldc_b, m0, n0, rows_here, wave, r16, q, park);
    } else {
        nt_epilogue<NB, EPI>(acc, mask_src, ldm_b, C, ldc_b, m0, n0, rows_here, wave, r16, q, park);
    }
}

// ------------------------------------------------------------------------------------------------ gemm_nt, bf16 operands
// Inference-only forward (BASELINE configs[4]: "bf16 fwd / fp32 master weights"): the fp32 activations and the fp32 master
// weights are staged exactly as in gemm_nt_dma_kernel (fp32 tiles in LDS), rounded to bf16 when a lane builds its MFMA
// operands, multiplied by v_mfma_f32_16x16x32_bf16 and accumulated / biased / activated in fp32.  Selected by
// rlppo_set_inference_precision(1) for the rollout entry points only; rlppo_ppo_minibatch never uses it.
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
    // 96-wide tile); the 32-wide tile (half a pass) stays at BK = 32 unless K is only a multiple of 16 [r3: a first layer whose
    // input is padded to 16], where it runs BK = 16 with its one ragged pass
    constexpr int BKT = (NB * 16) % 32 == 0 && NB * 16 >= 64 ? 16 : 32;
    if (BKT == 32 && K % 32 != 0) {
        if (epi == EPI_BIAS_RELU || epi == EPI_BIAS || epi == EPI_BIAS_TANH || epi == EPI_MASK) {
#define NT16(E)                                                                                                         \
    case E:                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_dma_kernel<NB, E, 16>), grid, dim3(256), 0, st, A, lda_b, B, ldb_b, bias, mask_src,  \
                           ldm_b, C, ldc_b, M, K);                                                                     \
        break;
            switch (epi) { NT16(EPI_BIAS) NT16(EPI_BIAS_RELU) NT16(EPI_BIAS_TANH) NT16(EPI_MASK) }
#undef NT16
            RLPPO_LAUNCH_CHECK();
            return 0;
        }
    }
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

// ------------------------------------------------------------------------------------------------ gemm_nt for a FEW rows [r5]
// The layer chain at rollout sizes (8 ... a few hundred rows: the Gaussian / multi-discrete heads, networks wider than the one-launch
// kernel's 256): gemm_nt_dma_kernel gives a 128-row tile to a workgroup, so 16 rows of a 512 -> 512 layer are TWO workgroups that
// multiply 7/8 padding -- ~28 us per layer, 172 us for a 5-layer call.  Here one WAVE owns one 16 x 16 output block (16 rows x 16
// columns) over the whole K and a workgroup is four of them: 128 waves for 16 x 512 outputs, each a dependent chain of K/4 MFMAs
// (1.7 us at K = 512) fed straight from L2 into registers, 8 K-tiles of both operands requested ahead.  The arithmetic is
// gemm_nt_dma_kernel's, operation for operation (accumulators start from the bias; K-tile kt, MFMA s contracts k = 16 kt + 4 q + s;
// relu as v_med3): outputs are BIT-identical (tests/test_gpu_kernels.py::test_gemm_nt_skinny).
namespace {
constexpr int SK_AHEAD = 8;          // K-tiles of 16 in flight per operand
constexpr int64_t SK_MAX_ROWS = 1024;  // beyond that the 128-row tiles fill the chip as well, at half the operand traffic
int g_nt_skinny = 1;                 // rlppo_dbg_set(40, 0/1)
}  // namespace
void set_nt_skinny(int on) { g_nt_skinny = on; }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                             const float *__restrict__ bias, float *__restrict__ C, int64_t ldc, int64_t M, int N,
                                                             int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int n0 = ((int)blockIdx.x * 4 + wave) * 16;
    if (n0 >= N) return;  // (whole waves; the kernel has no barrier)
    const int64_t m0 = (int64_t)blockIdx.y * 16;
    const int64_t arow = m0 + r16 < M ? m0 + r16 : M - 1;  // rows past M multiply a valid row; their products are not stored
    const float *ap = A + arow * lda + q * 4;
    const float *bp = B + (int64_t)(n0 + r16) * ldb + q * 4;
    f32x4 acc = *reinterpret_cast<const f32x4 *>(bias + n0 + q * 4);
    const int nk = K / 16;
    f32x4 fa[2][SK_AHEAD], fb[2][SK_AHEAD];
    auto fetch = [&](int buf, int kt0) {
#pragma unroll
        for (int i = 0; i < SK_AHEAD; ++i) {
            // (unconditional: a K-tile past the end is the last one again, its products are skipped -- a branch per request would
            // make the compiler wait for everything in flight before the first product)
            const int kt = kt0 + i < nk ? kt0 + i : nk - 1;
            fa[buf][i] = *reinterpret_cast<const f32x4 *>(ap + kt * 16);
            fb[buf][i] = *reinterpret_cast<const f32x4 *>(bp + kt * 16);
        }
    };
    fetch(0, 0);
    for (int kt0 = 0; kt0 < nk; kt0 += 2 * SK_AHEAD) {
        fetch(1, kt0 + SK_AHEAD);
#pragma unroll
        for (int i = 0; i < SK_AHEAD; ++i)
            if (kt0 + i < nk) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = MFMA16(fb[0][i][s], fa[0][i][s], acc);
            }
        fetch(0, kt0 + 2 * SK_AHEAD);
#pragma unroll
        for (int i = 0; i < SK_AHEAD; ++i)
            if (kt0 + SK_AHEAD + i < nk) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = MFMA16(fb[1][i][s], fa[1][i][s], acc);
            }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (EPI == EPI_BIAS_RELU) acc[e] = relu1(acc[e]);
        if (EPI == EPI_BIAS_TANH) acc[e] = tanhf(acc[e]);
    }
    // lane (r16, q) holds C[row m0 + r16][columns n0 + 4 q .. + 3]
    if (m0 + r16 < M) *reinterpret_cast<f32x4 *>(C + (m0 + r16) * ldc + n0 + q * 4) = acc;
}

static int launch_gemm_nt_skinny(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                                 int64_t ldc, int64_t M, int N, int K, int epi) {
    dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 16));
#define SK(E)                                                                                                              \
    case E:                                                                                                                \
        hipLaunchKernelGGL((gemm_nt_skinny_kernel<E>), grid, dim3(256), 0, st, A, lda, B, ldb, bias, C, ldc, M, N, K);      \
        break;
    switch (epi) {
        SK(EPI_BIAS) SK(EPI_BIAS_RELU) SK(EPI_BIAS_TANH)
        default:
            set_error("gemm_nt (few rows): bad epilogue %d", epi);
            return RLPPO_ERR_ARG;
    }
#undef SK
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

// floats of workspace one hidden layer's ReLU bitmask needs (8 bytes per lane and 128 x 128 tile); 0 = width not supported
size_t nt_bits_floats(int64_t M, int N) {
    if (N % 128 != 0 || M <= 0) return 0;
    return (size_t)cdiv(M, SBM) * (size_t)(N / 128) * 256 * 2;
}
// The hidden-layer forward (epi = EPI_BIAS_RELU: writes `bits`) or the masked dX product (epi = EPI_MASK: reads `bits`
// instead of the activation) through gemm_nt_dma_kernel<8, ., 16, true>.  Returns -1 when that form does not apply
// (width not a multiple of 128, operands too wide for 32-bit tile offsets): the caller then uses launch_gemm_nt
// and, for the forward, must not hand the bitmask to the backward pass.
// rowtab != nullptr (forward only): row r of A is A[rowtab[r]] of a `src_rows`-row matrix -- the fused minibatch gather.
static int g_pair_interleave = 1;  // rlppo_dbg_set(33, 0/1): paired launches interleave the two products' tiles (y range) instead of stacking them (z)
void set_pair_interleave(int v) { g_pair_interleave = v; }
bool nt_gather_ok(int64_t lda, int64_t src_rows, int N, int K) {
    return N % 128 == 0 && K % 16 == 0 && lda * 4 < 16384 && lda % 4 == 0 && src_rows > 0 && src_rows < ((int64_t)1 << 32);
}
int launch_gemm_nt_bits(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int N, int K, int epi, unsigned long long *bits, const unsigned *rowtab,
                        int64_t src_rows, const NtAlt *alt, const NtDot *dot) {
    if (!bits || N % 128 != 0 || K % 16 != 0 || M <= 0) return -1;
    if (epi != EPI_BIAS_RELU && epi != EPI_MASK) return -1;
    const int64_t lim = (int64_t)1 << 31;
    if ((!rowtab && 129 * lda * 4 >= lim) || 129 * ldb * 4 >= lim || 129 * ldc * 4 >= lim) return -1;
    RLPPO_CHECK_ARG(!alt || (alt->A && alt->B && alt->C && alt->bits), "gemm_nt: incomplete second operand set");
    dim3 grid((unsigned)cdiv(M, SBM), (unsigned)(N / 128), alt ? 2u : 1u);
    NtAlt second = alt ? *alt : NtAlt{};
    if (alt && g_pair_interleave) {
        grid.y *= 2;
        grid.z = 1;
        second.interleave = 1;
    }
    const NtDot d0 = dot ? dot[0] : NtDot{}, d1 = dot && alt ? dot[1] : NtDot{};
    RLPPO_CHECK_ARG(!dot || (epi == EPI_BIAS_RELU && N / 128 <= 2 && (!d0.w || (d0.out && d0.b)) && (!d1.w || (d1.out && d1.b))),
                    "gemm_nt: the folded one-output head needs the ReLU forward form and at most two column tiles");
    const unsigned la = (unsigned)(lda * 4), lb = (unsigned)(ldb * 4), lc = (unsigned)(ldc * 4);
    if (rowtab) {
        RLPPO_CHECK_ARG(epi == EPI_BIAS_RELU && nt_gather_ok(lda, src_rows, N, K), "gemm_nt (gathered rows): unsupported shape");
        hipLaunchKernelGGL((gemm_nt_dma_kernel<8, EPI_BIAS_RELU, 16, true, true>), grid, dim3(256), 0, st, A, la, B, lb, bias, nullptr,
                           0u, C, lc, M, K, bits, rowtab, (unsigned)src_rows, second, d0, d1);
    } else if (epi == EPI_BIAS_RELU)
        hipLaunchKernelGGL((gemm_nt_dma_kernel<8, EPI_BIAS_RELU, 16, true>), grid, dim3(256), 0, st, A, la, B, lb, bias, nullptr,
                           0u, C, lc, M, K, bits, nullptr, 0u, second, d0, d1);
    else
        hipLaunchKernelGGL((gemm_nt_dma_kernel<8, EPI_MASK, 16, true>), grid, dim3(256), 0, st, A, la, B, lb, nullptr, nullptr, 0u,
                           C, lc, M, K, bits, nullptr, 0u, second);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_nt(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                   const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int N, int K, int epi,
                   int bf16_operands) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(K > 0 && K % 16 == 0, "gemm_nt: K=%d must be a positive multiple of 16", K);
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
    // (the bf16-operand inference kernel steps K by 32: a layer whose K is only a multiple of 16 -- a first layer padded to 16 --
    // keeps the fp32 kernel, i.e. is computed MORE precisely than the mode asks)
    if (bf16_operands && epi != EPI_MASK && K % 32 == 0) return launch_gemm_nt_bf16(st, A, lda, B, ldb, bias, C, ldc, M, N, nb, K, epi);
    if (g_nt_skinny && M <= SK_MAX_ROWS && epi != EPI_MASK && !bf16_operands) return launch_gemm_nt_skinny(st, A, lda, B, ldb, bias, C, ldc, M, N, K, epi);
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
// 110-256 KB serialise in the memory-side atomic units).  The workgroup stores its partial tile with coalesced 16-byte
// stores per lane into partial[split][tile][(i*NJ+j)*256 + tid] and tn_reduce_kernel sums the splits afterwards: the
// sum order is fixed, so the weight gradients are bit-reproducible from run to run.
//
// Tile geometry <NI, NJ, WN> [r3]: the 4 waves form a WN x (4/WN) grid, a wave owns NI x NJ blocks of 16 x 16, so the
// workgroup's output tile is (16 NI WN) rows of dW x (16 NJ 4/WN) columns.  <4,4,2> = 128 x 128 (hidden layers); <3,4,2> =
// 96 x 128 for the 90-action head (was 128: a quarter of the MFMAs multiplied zero padding); <2,7,4> = 128 x 112 for the first
// layer of a 107-wide observation (was 128).  The stage images keep 128-float rows whatever the tile.
//
// GATHER [r3]: X is the experience buffer's state matrix and sample m of the contraction is X[rowtab[m]]: the X
// descriptor is a structured buffer addressed by a row index per lane and piece; the indices of stage s + 2 are fetched (no
// vector ALU: descriptor + lane-constant offset + scalar offset) right after the DMA of stage s + 1 has been issued.
struct TnGeom {
    int ni, nj, wn;
    __host__ __device__ int bnt() const { return wn * ni * 16; }
    __host__ __device__ int bkx() const { return (4 / wn) * nj * 16; }
    __host__ __device__ int tile_floats() const { return ni * nj * 1024; }
};
static TnGeom tn_geometry(int out, int in) {
    if (in > 96 && in <= 112) return TnGeom{2, 7, 4};
    if (out > 64 && out <= 96) return TnGeom{3, 4, 2};
    return TnGeom{4, 4, 2};
}

// One (split, tile) work item of a weight-gradient product: the body shared by the per-product launch (gemm_tn_dma_kernel) and the
// grouped launch (gemm_tn_group_kernel, all products of a pass in one grid).  `lds`: the workgroup's stage buffers (the kernels own
// the allocation: a static array per instantiation of this body would add up).  (bx, by, bz) = (row tile of dW, column tile, split);
// tiles_x / tiles_y / splits_z = the product's tile and split counts (they place the partial tile and the db column sums in `partial`).
constexpr int TN_LDS_FLOATS = 2 * 2 * 32 * 128;
template <int TMT, int NI, int NJ, int WN, bool GATHER>
__device__ __forceinline__ void tn_tile(float *lds, const float *__restrict__ dY, unsigned ldy_b, int ny_valid,
                                        const float *__restrict__ X, unsigned ldx_b, int kx_valid, bool with_db, int out, int64_t M,
                                        int rows_per_wg, float *__restrict__ partial, const unsigned *__restrict__ rowtab,
                                        unsigned src_rows, int bx, int by, int bz, int tiles_x, int tiles_y, int splits_z) {
    constexpr int WK = 4 / WN;
    constexpr int BNT = WN * NI * 16, BKX = WK * NJ * 16, TILE_F = NI * NJ * 1024;
    constexpr int PPW = TMT / 8;  // 1 KiB pieces per wave, per operand and stage
    static_assert(2 * 2 * TMT * 128 <= TN_LDS_FLOATS && 8 * 128 <= TN_LDS_FLOATS, "stage buffers");
    float *Ys = lds;                 // [2][TMT][128]
    float *Xs = lds + 2 * TMT * 128;  // [2][TMT][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int r16 = lane & 15, q = lane >> 4;
    const int wn = wave / WK, wk = wave % WK;
    const int n0 = bx * BNT, k0 = by * BKX;
    const int64_t mbeg = (int64_t)bz * rows_per_wg;
    const int rows = (int)((M - mbeg) < rows_per_wg ? (M - mbeg) : rows_per_wg);  // >= 1
    const int steps = (rows + TMT - 1) / TMT;
    const int rem = rows - (steps - 1) * TMT;  // rows of the last stage, 1..TMT
    const int ny_here = (ny_valid - n0) < BNT ? (ny_valid - n0) : BNT;
    const int kx_here = (kx_valid - k0) < BKX ? (kx_valid - k0) : BKX;

    const __amdgpu_buffer_rsrc_t y_rs = make_rsrc(reinterpret_cast<const char *>(dY) + mbeg * ldy_b + (int64_t)n0 * 4,
                                                  (unsigned)(rows - 1) * ldy_b + (unsigned)ny_here * 4);
    const __amdgpu_buffer_rsrc_t x_rs =
        GATHER ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), (short)ldx_b, src_rows, 0x00020000)
               : make_rsrc(reinterpret_cast<const char *>(X) + mbeg * ldx_b + (int64_t)k0 * 4,
                           (unsigned)(rows - 1) * ldx_b + (unsigned)kx_here * 4);
    // DMA lane map: piece = rows 2 (wave + 4 i) + (lane >> 5); physical chunk lane & 31 holds logical chunk ^ ((row & 3) << 2)
    const int row_l = 2 * wave + (lane >> 5);
    const int lch = (lane & 31) ^ ((row_l & 3) << 2);
    const bool x_col = lch * 4 < kx_here;
    const unsigned y_off = (lch * 4 < ny_here) ? (unsigned)row_l * ldy_b + lch * 16 : OOR;
    const unsigned x_off = GATHER ? (unsigned)(k0 * 4 + lch * 16) : (x_col ? (unsigned)row_l * ldx_b + lch * 16 : OOR);
    const unsigned y_row8 = 8u * ldy_b, x_row8 = 8u * ldx_b, y_stage = TMT * ldy_b, x_stage = TMT * ldx_b;  // uniform
    // GATHER: buffer rows of this lane's PPW pieces of the stage about to be issued (entries past the split read as 0: a valid
    // row whose dY partner rows the range check / clear_tail zero)
    unsigned x_row[4];  // (a bound that depends on a template parameter, captured by the lambdas below, loses the host-side launch stub: hipcc 7.2)
    static_assert(PPW <= 4, "x_row");
    const __amdgpu_buffer_rsrc_t t_rs = make_rsrc(GATHER ? rowtab + mbeg : nullptr, GATHER ? (unsigned)rows * 4 : 0u);
    auto load_rows = [&](int stage) {
        if (GATHER) {
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                x_row[i] = __builtin_amdgcn_raw_buffer_load_b32(t_rs, (unsigned)row_l * 4, (unsigned)(stage * TMT + 8 * i) * 4, 0);
        }
    };

    auto issue_stage = [&](int buf, int stage) {
        float *Yd = Ys + (buf * TMT + 2 * wave_u) * 128, *Xd = Xs + (buf * TMT + 2 * wave_u) * 128;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rs, Yd + 8 * i * 128, 16, y_off, (unsigned)stage * y_stage + i * y_row8, 0, 0);
            if (GATHER) {
                if (x_col) __builtin_amdgcn_struct_ptr_buffer_load_lds(x_rs, Xd + 8 * i * 128, 16, x_row[i], x_off, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, Xd + 8 * i * 128, 16, x_off, (unsigned)stage * x_stage + i * x_row8, 0, 0);
            }
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

    // [r5] TWO accumulation chains per output block, taking the 16-row halves of every stage in turn: a split's rows are summed as
    // two fp32 chains of half the length (added once, at the end), so the rounding of a split of R rows is that of chains of R / 2
    // -- the grouped launch's splits are 2,700-7,300 rows long where rounds 1-4 ran at most 4,096 -- and the MFMA stream has two
    // independent chains per block to interleave.  64 more VGPRs; the launch bound (two workgroups per CU: LDS) is unchanged.
    f32x4 acc[NI][NJ], acc2[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bs4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_db = with_db && by == 0;

    load_rows(0);
    if (GATHER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_stage(0, 0);
    if (steps > 1) load_rows(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (steps == 1 && rem < TMT) clear_tail(0);
    __syncthreads();
    // fragment addresses: element (m, col) lives at m*128 + (col ^ ((m & 3) << 4)); m & 3 == q for every fragment read
    int fy[NI], fx[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = wn * NI + i;
        fy[i] = q * 128 + (((g & ~3) | ((g & 3) ^ q)) << 4) + r16;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int g = wk * NJ + j;
        fx[j] = q * 128 + (((g & ~3) | ((g & 3) ^ q)) << 4) + r16;
    }
    const int db_off = zr * 128 + zc;  // db partial sums: rows zr + 8 i, physical chunk tid & 31
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        const bool more = (st + 1) < steps;
        if (more) {
            issue_stage(cur ^ 1, st + 1);
            if ((st + 2) < steps) load_rows(st + 2);
        }
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
                float fa[NI], fb[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[i] = Yc[m * 128 + fy[i]];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = Xc[m * 128 + fx[j]];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (c & 1) acc2[i][j] = MFMA16(fa[i], fb[j], acc2[i][j]);
                        else acc[i][j] = MFMA16(fa[i], fb[j], acc[i][j]);
                    }
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
        if (tid < BNT && (n0 + tid) < out) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) sum += red[r * 128 + tid];
            // column sums of this split: partial_db[split][n tile][128], behind the tile partials
            partial[(size_t)splits_z * tiles_y * tiles_x * TILE_F + ((size_t)bz * tiles_x + bx) * 128 + tid] = sum;
        }
    }
    // The partial-tile stores are the LAST instructions of the wave: 16-byte buffer stores whose data registers are written
    // again soon afterwards can pick up the new values (the hazard of section 5 / tests/test_gpu_stress.py); here the
    // accumulators are never touched after them.
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] += acc2[i][j];
    __builtin_amdgcn_sched_barrier(0);
    {
        const size_t tile_id = (size_t)bz * (tiles_x * tiles_y) + (size_t)by * tiles_x + bx;
        const __amdgpu_buffer_rsrc_t p_rs = make_rsrc(partial + tile_id * TILE_F, TILE_F * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) stb(p_rs, (unsigned)tid * 16, (unsigned)(i * NJ + j) * 4096, acc[i][j]);
    }
}

template <int TMT, int NI, int NJ, int WN, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_tn_dma_kernel(const float *__restrict__ dY, unsigned ldy_b,
                                                                              int ny_valid, const float *__restrict__ X,
                                                                              unsigned ldx_b, int kx_valid,
                                                                              bool with_db, int out, int in, int64_t M,
                                                                              int rows_per_wg, float *__restrict__ partial,
                                                                              const unsigned *__restrict__ rowtab,
                                                                              unsigned src_rows, TnAlt alt) {
    __shared__ __attribute__((aligned(16))) float lds[TN_LDS_FLOATS];
    // [r3] two products of the same shape in one launch: the upper half of the z range takes (dY, X, partial) from `alt`
    const int splits_z = alt.dY ? gridDim.z / 2 : gridDim.z;
    int zloc = blockIdx.z;
    bool second = alt.dY && !alt.interleave && zloc >= splits_z;
    if (second) zloc -= splits_z;
    // interleaved pairing (splits a multiple of 8): id -> (product, tile, split) in groups of 8 splits x 2 T tiles
    int il_tile = -1, il_split = 0;
    if (alt.dY && alt.interleave) {
        const int T = gridDim.x * gridDim.y;
        const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, g = id % (16 * T);
        int t2 = g >> 3;
        second = t2 >= T;
        if (second) t2 -= T;
        il_tile = t2;
        il_split = (id / (16 * T)) * 8 + (g & 7);
    }
    if (second) {
        dY = alt.dY;
        if (!GATHER) X = alt.X;
        partial = alt.partial;
    }
    // XCD-aware order (see xcd_tile): the output tiles of one row split read the same dY / X rows, so they are given ids
    // that land on the same XCD back to back: id -> tile = (id % (8 T)) / 8, split = 8 (id / (8 T)) + id % 8
    int bx = blockIdx.x, by = blockIdx.y, bz = zloc;
    {
        const int T = gridDim.x * gridDim.y;
        if (il_tile >= 0) {
            bz = il_split;
            bx = il_tile % gridDim.x;
            by = il_tile / gridDim.x;
        } else if ((splits_z & 7) == 0 && T > 1) {
            const int id = (zloc * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, g = id % (8 * T);
            const int tile = g >> 3;
            bz = (id / (8 * T)) * 8 + (g & 7);
            bx = tile % gridDim.x;
            by = tile / gridDim.x;
        }
    }
    tn_tile<TMT, NI, NJ, WN, GATHER>(lds, dY, ldy_b, ny_valid, X, ldx_b, kx_valid, with_db, out, M, rows_per_wg, partial, rowtab, src_rows,
                                     bx, by, bz, (int)gridDim.x, (int)gridDim.y, splits_z);
}


// Sums the partial tiles of gemm_tn_dma_kernel over the splits and adds the result into dW[out][in].
// Block = 64 consecutive 16-byte elements of one tile x 4 split lanes (one wave each: 1 KiB coalesced per load, 8 loads
// in flight); the four partial sums meet in LDS.  The final add is an atomic only so that launches of different
// minibatches that share dW stay safe; there is exactly one per element and launch.  geo: the producing kernel's tile
// geometry (a tile is geo.ni * geo.nj * 256 such elements: gridDim.x = geo.ni * geo.nj * 4 blocks per tile).
__device__ __forceinline__ void tn_reduce_tile(const float *__restrict__ partial, int splits, int tiles_x, int tiles,
                                               float *__restrict__ dW, float *__restrict__ db, int out, int in, TnGeom geo, int tile,
                                               int bxi, int nbx, float (*red)[64][4], float *dbh) {
    const int e64 = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int elem4 = bxi * 64 + e64;  // 16-byte element of the tile: (i*NJ+j)*256 + tid
    const int tile4 = geo.tile_floats() / 4;
    const f32x4 *base = reinterpret_cast<const f32x4 *>(partial) + (size_t)tile * tile4 + elem4;
    const size_t stride = (size_t)tiles * tile4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = sl;
    for (; sp + 28 < splits; sp += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + (size_t)(sp + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sp + 12 < splits; sp += 16) {  // (a grouped launch has ~24 splits per product: keep four loads in flight there too)
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(base + (size_t)(sp + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    for (; sp < splits; sp += 4) acc += __builtin_nontemporal_load(base + (size_t)sp * stride);
    if (sl > 0) *reinterpret_cast<f32x4 *>(&red[sl - 1][e64][0]) = acc;
    // bias gradient (one block per n tile): 128 columns x 2 halves of the splits, 16 loads in flight per thread, every
    // partial combined in a fixed order
    const int bnt = geo.bnt(), bkx = geo.bkx();
    const bool db_block = db && bxi == nbx - 1 && tile < tiles_x;
    float sdb = 0.f;
    if (db_block) {
        const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
        const float *pdb = partial + (size_t)splits * tiles * geo.tile_floats() + (size_t)tile * 128 + c;
        const size_t dstride = (size_t)tiles_x * 128;
        const int s_lo = half ? (splits + 1) / 2 : 0, s_hi = half ? splits : (splits + 1) / 2;
        float a16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a16[u] = 0.f;
        int s2 = s_lo;
        if (c < bnt) {  // (columns past the tile's width were never written)
            for (; s2 + 15 < s_hi; s2 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = pdb[(size_t)(s2 + u) * dstride];
#pragma unroll
                for (int u = 0; u < 16; ++u) a16[u] += v[u];
            }
            for (int u = 0; s2 < s_hi; ++s2, ++u) a16[u] += pdb[(size_t)s2 * dstride];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) sdb += a16[u];
        if (half) dbh[c] = sdb;
    }
    __syncthreads();
    if (db_block && threadIdx.x < bnt && tile * bnt + (int)threadIdx.x < out)
        atomicAdd(db + tile * bnt + threadIdx.x, sdb + dbh[threadIdx.x]);
    if (sl > 0) return;
#pragma unroll
    for (int r = 0; r < 3; ++r) acc += *reinterpret_cast<const f32x4 *>(&red[r][e64][0]);
    const int ij = elem4 >> 8, t = elem4 & 255;
    const int i = ij / geo.nj, j = ij % geo.nj, wave = t >> 6, lane = t & 63;
    const int wkc = 4 / geo.wn, wn = wave / wkc, wk = wave % wkc;
    const int n0 = (tile % tiles_x) * bnt, k0 = (tile / tiles_x) * bkx;
    const int k = k0 + (wk * geo.nj + j) * 16 + (lane & 15);
    const int nb = n0 + (wn * geo.ni + i) * 16 + (lane >> 4) * 4;
    if (k < in) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (nb + e < out) atomicAdd(dW + (size_t)(nb + e) * in + k, acc[e]);
    }
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float *__restrict__ partial, int splits, int tiles_x, int tiles,
                                                        float *__restrict__ dW, float *__restrict__ db, int out, int in,
                                                        TnGeom geo, TnRedAlt alt) {
    if (blockIdx.z) {  // the second product of a paired launch
        partial = alt.partial;
        dW = alt.dW;
        db = alt.db;
    }
    __shared__ __attribute__((aligned(16))) float red[3][64][4];
    __shared__ float dbh[128];
    tn_reduce_tile(partial, splits, tiles_x, tiles, dW, db, out, in, geo, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, red, dbh);
}

// rows per workgroup: no atomic traffic to trade against, so simply two workgroups per CU
static int64_t tn_rows_for(int64_t tiles, int64_t M) {
    int64_t rows = tiles >= 2 ? 512 : 256;
    if (M < 64 * rows) rows = round_up(cdiv(M, 64) > 32 ? cdiv(M, 64) : 32, 32);
    // large M (fused minibatches): one round of workgroups (2 per CU) instead of more and more splits -- fewer partial tiles
    // to write and to reduce.  M = 524,288, us per launch at 512 / 1024 / 2048 / 4096 rows: hidden 567 / 527 / 505 / 494,
    // first layer 313 / 284 / 267 / 287, policy head 305 / 277 / 262 / 284
    const int64_t few = round_up(cdiv(M * tiles, 512), 32);
    return few > rows ? few : rows;
}
int64_t tn_partial_rows(int out, int in, int64_t M) {  // the 128 x 128-tile forms (the bf16 kernel's partials are always that)
    return tn_rows_for(cdiv(out, 128) * cdiv(in, 128), M);
}

size_t tn_partial_floats(int out, int in, int64_t M) {
    if (M <= 0) return 0;
    // the fp32 kernel's geometry for this shape ...
    const TnGeom g = tn_geometry(out, in);
    const size_t tx = (size_t)cdiv(out, g.bnt()), ty = (size_t)cdiv(in, g.bkx());
    const size_t sp = (size_t)cdiv(M, tn_rows_for((int64_t)(tx * ty), M));
    size_t need = sp * tx * ty * (size_t)g.tile_floats() + sp * tx * 128;  // tiles + db
    // ... and the bf16 kernels' 128 x 128 partials (the wide form may split finer): size for whichever needs more
    size_t splits = (size_t)cdiv(M, tn_partial_rows(out, in, M));
    const int pout = (int)round_up(out, 128), pin = (int)round_up(in, 128);
    if (pout % 256 == 0 && pin % 256 == 0) {
        const size_t w = (size_t)cdiv(M, tn_partial_rows_wide(pout, pin, M));
        splits = w > splits ? w : splits;
    }
    const size_t b16 = splits * (size_t)(cdiv(out, 128) * cdiv(in, 128)) * (128 * 128) + splits * (size_t)cdiv(out, 128) * 128;
    return need > b16 ? need : b16;
}

// dW[out][in] += dY^T . X, db[out] += colsum(dY) through partial tiles in `ws` (>= tn_partial_floats floats) + a reduction.
// rowtab != nullptr: sample m of the contraction is X[rowtab[m]] of a `src_rows`-row matrix (the fused minibatch gather).
bool tn_gather_ok(int64_t ldx, int64_t src_rows) { return ldx * 4 < 16384 && ldx % 4 == 0 && src_rows > 0 && src_rows < ((int64_t)1 << 32); }
int launch_gemm_tn(hipStream_t st, const float *dY, int64_t ldy, int ny_valid, const float *X, int64_t ldx, int kx_valid,
                   float *dW, float *db, int out, int in, int64_t M, float *ws, size_t ws_floats, const unsigned *rowtab,
                   int64_t src_rows, const TnPair *pair) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(ny_valid % 4 == 0 && kx_valid % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ny_valid <= ldy &&
                        kx_valid <= ldx && out <= ny_valid && in <= kx_valid,
                    "gemm_tn: bad shapes ny=%d kx=%d ldy=%ld ldx=%ld out=%d in=%d", ny_valid, kx_valid, (long)ldy,
                    (long)ldx, out, in);
    if (!ws || ws_floats < tn_partial_floats(out, in, M)) {
        set_error("gemm_tn: workspace %zu < %zu floats", ws ? ws_floats : (size_t)0, tn_partial_floats(out, in, M));
        return RLPPO_ERR_WORKSPACE;
    }
    const TnGeom g = tn_geometry(out, in);
    const int tiles_x = (int)cdiv(out, g.bnt()), tiles_y = (int)cdiv(in, g.bkx());
    const int rows_per_wg = (int)tn_rows_for((int64_t)tiles_x * tiles_y, M);
    const int splits = (int)cdiv(M, rows_per_wg);
    const int64_t lim = (int64_t)1 << 30;
    RLPPO_CHECK_ARG((rows_per_wg + TM) * ldy * 4 < lim && (rowtab || (rows_per_wg + TM) * ldx * 4 < lim),
                    "gemm_tn: a leading dimension is too wide for 32-bit tile offsets");
    RLPPO_CHECK_ARG(!rowtab || tn_gather_ok(ldx, src_rows), "gemm_tn (gathered rows): unsupported row stride %ld", (long)ldx);
    RLPPO_CHECK_ARG(!pair || (pair->dY && (rowtab || pair->X) && pair->dW && pair->ws && (pair->db != nullptr) == (db != nullptr)),
                    "gemm_tn: incomplete second operand set");
    dim3 grid((unsigned)tiles_x, (unsigned)tiles_y, (unsigned)(pair ? 2 * splits : splits));
    TnAlt alt{};
    if (pair) alt = TnAlt{pair->dY, pair->X, pair->ws, (g_pair_interleave && splits % 8 == 0) ? 1 : 0};
#define TN(NI_, NJ_, WN_, G_)                                                                                              \
    hipLaunchKernelGGL((gemm_tn_dma_kernel<TM, NI_, NJ_, WN_, G_>), grid, dim3(256), 0, st, dY, (unsigned)(ldy * 4), ny_valid, X, \
                       (unsigned)(ldx * 4), kx_valid, db != nullptr, out, in, M, rows_per_wg, ws, rowtab, (unsigned)src_rows, alt)
    if (g.nj == 7) {
        if (rowtab) TN(2, 7, 4, true); else TN(2, 7, 4, false);
    } else if (g.ni == 3) {
        RLPPO_CHECK_ARG(!rowtab, "gemm_tn (gathered rows): the 96-row tile has no gathered form");
        TN(3, 4, 2, false);
    } else {
        if (rowtab) TN(4, 4, 2, true); else TN(4, 4, 2, false);
    }
#undef TN
    RLPPO_LAUNCH_CHECK();
    TnRedAlt ralt{};
    if (pair) ralt = TnRedAlt{pair->ws, pair->dW, pair->db};
    return launch_tn_reduce(st, ws, splits, tiles_x, tiles_y, dW, db, out, in, g.ni, g.nj, g.wn, pair ? &ralt : nullptr);
}
int launch_tn_reduce(hipStream_t st, const float *partial, int splits, int tiles_x, int tiles_y, float *dW, float *db, int out, int in,
                     int ni, int nj, int wn, const TnRedAlt *alt) {
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(ni * nj * 4), (unsigned)(tiles_x * tiles_y), alt ? 2u : 1u), dim3(256), 0, st,
                       partial, splits, tiles_x, tiles_x * tiles_y, dW, db, out, in, TnGeom{ni, nj, wn}, alt ? *alt : TnRedAlt{});
    RLPPO_LAUNCH_CHECK();
    return 0;
}
// ------------------------------------------------------------------------------------------------ [r5] grouped weight gradients
// Every dW / db product of a pass in ONE launch.  Why: a per-layer launch at the 65,536 rows of one rank of an 8-rank job wrote 32 MB
// of partial tiles (128 splits x 4 tiles x 64 KB) and was followed by a reduction launch that read them back -- 7 such pairs per
// optimiser step, 232 us of 1155 in reductions alone, the dW products at 0.50-0.72 of the MFMA peak against 0.89 for the same kernel
// at 524,288 rows.  The products of a backward pass have no consumer before the optimiser step, so they can all run at its end: one
// grid of (about) two workgroups per CU, each workgroup one (product, output tile, row split) item, the splits of a product sized
// so that every item is the same number of MFMA blocks (rows x NI x NJ).  A split is then thousands of rows long (the K loop's
// prologue / epilogue amortised as in the large launch), the partial tiles of the WHOLE pass are 512 x 64 KB = 32 MB, and one
// reduction launch sums them -- in split order, so the gradient stays bit-reproducible.
struct TnWork {  // device-side view of one product
    const float *dY, *X;
    float *partial;
    const unsigned *rowtab;
    unsigned ldy_b, ldx_b, src_rows;
    int ny_valid, kx_valid, out, rows_per_wg, splits, tiles_x, tiles_y, with_db, geom, wg_begin;
};
struct TnGroupArgs {
    int n;
    int64_t M;
    TnWork w[TN_GROUP_MAX];
};
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(TnGroupArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[TN_LDS_FLOATS];
    const int id = blockIdx.x;
    int p = 0;
    for (int i = 1; i < g.n; ++i)
        if (id >= g.w[i].wg_begin) p = i;  // (uniform: scalar loads from the kernel arguments)
    const TnWork &w = g.w[p];
    const int loc = id - w.wg_begin, T = w.tiles_x * w.tiles_y, S8 = w.splits & ~7;
    // XCD-aware order inside a product (its first id is a multiple of 8): whole groups of 8 splits put the T tiles of a split on
    // ONE XCD (they read the same dY / X rows), id -> tile = (id % (8 T)) / 8, split = 8 (id / (8 T)) + id % 8; the splits left
    // over are laid out tile by tile
    int tile, bz;
    if (loc < S8 * T) {
        const int gq = loc % (8 * T);
        tile = gq >> 3;
        bz = (loc / (8 * T)) * 8 + (gq & 7);
    } else {
        const int r = loc - S8 * T, R = w.splits - S8;
        if (R <= 0 || r >= R * T) return;  // padding up to the next product's first id
        tile = r / R;
        bz = S8 + r % R;
    }
    const int bx = tile % w.tiles_x, by = tile / w.tiles_x;
#define TN_ITEM(NI_, NJ_, WN_, G_)                                                                                                     \
    tn_tile<TM, NI_, NJ_, WN_, G_>(lds, w.dY, w.ldy_b, w.ny_valid, w.X, w.ldx_b, w.kx_valid, w.with_db != 0, w.out, g.M, w.rows_per_wg, \
                                   w.partial, w.rowtab, w.src_rows, bx, by, bz, w.tiles_x, w.tiles_y, w.splits)
    switch (w.geom) {
        case 0: TN_ITEM(4, 4, 2, false); break;
        case 1: TN_ITEM(4, 4, 2, true); break;
        case 2: TN_ITEM(3, 4, 2, false); break;
        case 3: TN_ITEM(2, 7, 4, false); break;
        default: TN_ITEM(2, 7, 4, true); break;
    }
#undef TN_ITEM
}

struct TnRedWork {
    const float *partial;
    float *dW, *db;
    int splits, tiles_x, tiles, out, in, tile_begin;
    TnGeom geo;
};
struct TnRedGroupArgs {
    int n;
    TnRedWork w[TN_GROUP_MAX];
};
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(TnRedGroupArgs g) {
    __shared__ __attribute__((aligned(16))) float red[3][64][4];
    __shared__ float dbh[128];
    const int t = blockIdx.y;
    int p = 0;
    for (int i = 1; i < g.n; ++i)
        if (t >= g.w[i].tile_begin) p = i;
    const TnRedWork &w = g.w[p];
    const int nbx = w.geo.ni * w.geo.nj * 4;
    if ((int)blockIdx.x >= nbx) return;
    tn_reduce_tile(w.partial, w.splits, w.tiles_x, w.tiles, w.dW, w.db, w.out, w.in, w.geo, t - w.tile_begin, (int)blockIdx.x, nbx, red, dbh);
}

// The plan of a grouped launch: splits per product such that every workgroup multiplies (about) the same number of 16 x 16 x 4
// blocks and the grid is at most `budget` workgroups (one round at two per CU).  Depends on the SET of shapes and M only.
static int g_tn_group_budget = 0;  // rlppo_dbg_set(38, n): 0 = two per CU
constexpr int TN_MAX_CHAIN = 8192;  // longest row split of a grouped launch (two interleaved fp32 accumulation chains of half that: tn_tile)
void set_tn_group_budget(int v) { g_tn_group_budget = v; }
struct TnPlanItem {
    TnGeom g;
    int tiles_x, tiles_y, splits, rows_per_wg, wg_begin;
    size_t partial_off, floats;
};
static int tn_group_plan(const int *outs, const int *ins, int n, int64_t M, int budget, TnPlanItem *it, int *total_wgs, size_t *total_floats) {
    for (int p = 0; p < n; ++p) {
        it[p].g = tn_geometry(outs[p], ins[p]);
        it[p].tiles_x = (int)cdiv(outs[p], it[p].g.bnt());
        it[p].tiles_y = (int)cdiv(ins[p], it[p].g.bkx());
    }
    const int64_t max_splits = cdiv(M, 32);
    auto fill = [&](double t) {  // t: MFMA blocks x rows per workgroup
        int64_t wgs = 0;
        for (int p = 0; p < n; ++p) {
            const int blocks = it[p].g.ni * it[p].g.nj;
            int64_t rows = (int64_t)(t / blocks);
            rows = round_up(rows < 32 ? 32 : rows, 32);
            if (g_tn_group_budget <= 0 && rows > TN_MAX_CHAIN) rows = TN_MAX_CHAIN;  // (a product of few MFMA blocks per row is not given longer chains for balance's sake)
            int64_t sp = cdiv(M, rows);
            if (sp > max_splits) sp = max_splits;
            rows = round_up(cdiv(M, sp), 32);
            it[p].rows_per_wg = (int)rows;
            it[p].splits = (int)cdiv(M, rows);
            wgs += round_up((int64_t)it[p].splits * it[p].tiles_x * it[p].tiles_y, 8);
        }
        return wgs;
    };
    // One round of the grid when the splits stay short enough; else 2, 3, ... whole rounds: a split is ONE fp32 accumulation chain
    // per output element, and its rounding error grows with the square root of its length -- the per-layer launches of rounds 1-4
    // never ran more than 4096 rows per split, TN_MAX_CHAIN keeps the grouped launch within sqrt(2) of that at any M (524,288 rows:
    // three rounds of 7,296-row splits).  (An explicit budget -- rlppo_dbg_set(38), tests -- is taken as it is.)
    // (from TWO rounds of the grid: at every size measured the two-round plan beats the one-round plan by 2-3 % -- workgroups of the second
    // round start as those of the first finish, one by one, so stores, stage fills and MFMA streams of different workgroups overlap --
    // and more rounds than the cap on the split length needs cost 1-3 % each: tools/tn_group_budget_sweep.py, profiles/r05_tn_group_budget_sweep.txt)
    for (int rounds = g_tn_group_budget > 0 ? 1 : 2;; ++rounds) {
        const int64_t cap = (int64_t)budget * rounds;
        double lo = 32.0, hi = (double)round_up(M, 32) * 32.0;  // fill(hi): the fewest splits the cap on their length allows
        if (fill(lo) <= cap) break;                              // (hardly any rows: 32-row splits fit the grid)
        for (int i = 0; i < 60; ++i) {
            const double mid = 0.5 * (lo + hi);
            if (fill(mid) > cap) lo = mid; else hi = mid;
        }
        if (fill(hi) <= cap || g_tn_group_budget > 0 || rounds >= 256) break;  // (with the splits capped, large M needs more than one round)
    }
    int wg = 0;
    size_t off = 0;
    for (int p = 0; p < n; ++p) {
        it[p].wg_begin = wg;
        wg += (int)round_up((int64_t)it[p].splits * it[p].tiles_x * it[p].tiles_y, 8);
        it[p].partial_off = off;
        it[p].floats = (size_t)it[p].splits * it[p].tiles_x * it[p].tiles_y * (size_t)it[p].g.tile_floats() + (size_t)it[p].splits * it[p].tiles_x * 128;
        off += (it[p].floats + 3) / 4 * 4;
    }
    *total_wgs = wg;
    *total_floats = off;
    return 0;
}
static int tn_group_budget() {
    if (g_tn_group_budget > 0) return g_tn_group_budget;
    int cus = 0;
    if (device_cu_count(&cus) || cus <= 0) cus = 256;
    return 2 * cus;
}
size_t tn_group_floats(const int *outs, const int *ins, int n, int64_t M) {
    if (n <= 0 || M <= 0) return 0;
    size_t total = 0;
    for (int b = 0; b < n; b += TN_GROUP_MAX) {  // (more products than one launch carries: consecutive launches share the buffer)
        TnPlanItem it[TN_GROUP_MAX];
        int wgs = 0;
        size_t f = 0;
        const int m = n - b < TN_GROUP_MAX ? n - b : TN_GROUP_MAX;
        tn_group_plan(outs + b, ins + b, m, M, tn_group_budget(), it, &wgs, &f);
        total = f > total ? f : total;
    }
    return total;
}
int launch_gemm_tn_group(hipStream_t st, const TnProduct *prods, int n, int64_t M, float *ws, size_t ws_floats) {
    if (M <= 0 || n <= 0) return 0;
    for (int b = 0; b < n; b += TN_GROUP_MAX) {
        const int m = n - b < TN_GROUP_MAX ? n - b : TN_GROUP_MAX;
        const TnProduct *pr = prods + b;
        int outs[TN_GROUP_MAX], ins[TN_GROUP_MAX];
        for (int p = 0; p < m; ++p) {
            outs[p] = pr[p].out;
            ins[p] = pr[p].in;
        }
        TnPlanItem it[TN_GROUP_MAX];
        int wgs = 0;
        size_t floats = 0;
        tn_group_plan(outs, ins, m, M, tn_group_budget(), it, &wgs, &floats);
        if (!ws || ws_floats < floats) {
            set_error("gemm_tn_group: workspace %zu < %zu floats", ws ? ws_floats : (size_t)0, floats);
            return RLPPO_ERR_WORKSPACE;
        }
        TnGroupArgs ga{};
        TnRedGroupArgs ra{};
        ga.n = ra.n = m;
        ga.M = M;
        const int64_t lim = (int64_t)1 << 30;
        int tile_begin = 0;
        for (int p = 0; p < m; ++p) {
            const TnProduct &q = pr[p];
            RLPPO_CHECK_ARG(q.dY && q.X && q.dW && q.ny_valid % 4 == 0 && q.kx_valid % 4 == 0 && q.ldy % 4 == 0 && q.ldx % 4 == 0 &&
                                q.ny_valid <= q.ldy && q.kx_valid <= q.ldx && q.out <= q.ny_valid && q.in <= q.kx_valid,
                            "gemm_tn_group: bad shapes (product %d) ny=%d kx=%d ldy=%ld ldx=%ld out=%d in=%d", p, q.ny_valid, q.kx_valid,
                            (long)q.ldy, (long)q.ldx, q.out, q.in);
            RLPPO_CHECK_ARG(((int64_t)it[p].rows_per_wg + TM) * q.ldy * 4 < lim && (q.rowtab || ((int64_t)it[p].rows_per_wg + TM) * q.ldx * 4 < lim),
                            "gemm_tn_group: a leading dimension is too wide for 32-bit tile offsets");
            RLPPO_CHECK_ARG(!q.rowtab || tn_gather_ok(q.ldx, q.src_rows), "gemm_tn_group (gathered rows): unsupported row stride %ld", (long)q.ldx);
            RLPPO_CHECK_ARG(!(q.rowtab && it[p].g.ni == 3), "gemm_tn_group (gathered rows): the 96-row tile has no gathered form");
            TnWork &w = ga.w[p];
            w.dY = q.dY;
            w.X = q.X;
            w.partial = ws + it[p].partial_off;
            w.rowtab = q.rowtab;
            w.ldy_b = (unsigned)(q.ldy * 4);
            w.ldx_b = (unsigned)(q.ldx * 4);
            w.src_rows = (unsigned)q.src_rows;
            w.ny_valid = q.ny_valid;
            w.kx_valid = q.kx_valid;
            w.out = q.out;
            w.rows_per_wg = it[p].rows_per_wg;
            w.splits = it[p].splits;
            w.tiles_x = it[p].tiles_x;
            w.tiles_y = it[p].tiles_y;
            w.with_db = q.db != nullptr;
            w.geom = it[p].g.nj == 7 ? (q.rowtab ? 4 : 3) : it[p].g.ni == 3 ? 2 : (q.rowtab ? 1 : 0);
            w.wg_begin = it[p].wg_begin;
            TnRedWork &r = ra.w[p];
            r.partial = w.partial;
            r.dW = q.dW;
            r.db = q.db;
            r.splits = it[p].splits;
            r.tiles_x = it[p].tiles_x;
            r.tiles = it[p].tiles_x * it[p].tiles_y;
            r.out = q.out;
            r.in = q.in;
            r.tile_begin = tile_begin;
            r.geo = it[p].g;
            tile_begin += r.tiles;
        }
        hipLaunchKernelGGL(gemm_tn_group_kernel, dim3((unsigned)wgs), dim3(256), 0, st, ga);
        RLPPO_LAUNCH_CHECK();
        hipLaunchKernelGGL(tn_reduce_group_kernel, dim3(64u, (unsigned)tile_begin), dim3(256), 0, st, ra);
        RLPPO_LAUNCH_CHECK();
    }
    return 0;
}
}  // namespace rlppo
