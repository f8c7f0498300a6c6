// gemm.hip -- fp32 MFMA GEMM kernels for the MLP forward / backward (gfx950, wave64).
//
// Why fp32 MFMA: BASELINE.json asks for fp32 losses/grads within 1e-5 relative of the reference's CPU path.
// v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain (MI355X_MICROARCH.md "Matrix cores") at the fp32
// vector peak (157 TFLOP/s), so parity needs no error analysis and the VALU stays free for epilogues.
//
//   gemm_nt : C[M][N] = epi(A[M][K] . B[N][K]^T)  -- both operands contraction-contiguous.
//             forward:   A = activations, B = packed W[out][in]              (epi = bias / bias+relu / bias+tanh)
//             backward:  A = dY,          B = packed W^T[in][out]            (epi = relu mask of the saved activation)
//             The first layer reads its rows through an index vector (the minibatch gather of
//             experience_buffer.py:82-87 fused into the A-tile load).
//   gemm_tn : dW[N][K] += dY[M][N]^T . X[M][K], db[N] += colsum(dY) -- contraction over the row (sample) axis,
//             split over workgroups along M and accumulated with fp32 atomics into the flat gradient arena
//             (the reference accumulates minibatch gradients into .grad the same way, ppo_learner.py:179-180).
//
// Tiling (both): 256 threads = 4 waves, LDS double buffer, one barrier per K step, 2 workgroups per CU.
// MFMA operand trick: a lane loads 4 consecutive k with ONE ds_read_b128 and feeds them to 4 successive
// 16x16x4 MFMAs; hardware k-slot (lane>>4) then covers k = 4*(lane>>4)+s in step s -- a permutation of the
// contraction order applied identically to both operands, so the product is unchanged.
#include "common.hpp"

namespace rlppo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int BM = 128;  // rows per workgroup (gemm_nt)
constexpr int BK = 32;   // K step (floats) = 8 chunks of 16 B per LDS row

// LDS tile of R rows x 32 floats, 16-byte chunks XOR-swizzled by (row & 7): ds_read_b128 fragment reads
// (16 rows x 4 chunks per instruction) then touch 16 distinct 16-B slots -> conflict free.
__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 2); }

template <int NB, int EPI, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float *__restrict__ A, int64_t lda,
                                                          const int64_t *__restrict__ row_idx,
                                                          const float *__restrict__ B, int64_t ldb,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ mask_src, int64_t ld_mask,
                                                          float *__restrict__ C, int64_t ldc, int64_t M, int K) {
    constexpr int BN = NB * 16;
    constexpr int B_ITERS = BN / 32;  // rows of the B tile each thread stages (32 rows per pass)
    __shared__ __attribute__((aligned(16))) float lds[2 * BM * BK + 2 * BN * BK];
    float *As = lds;
    float *Bs = lds + 2 * BM * BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // staging assignment: chunk = tid & 7 (16 B), rows (tid >> 3) + 32 * i
    const int ld_chunk = tid & 7, ld_row = tid >> 3;
    const float *a_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m0 + ld_row + 32 * i;
        if (m >= M) m = M - 1;  // clamp: rows past M are computed but never stored
        int64_t src = GATHER ? row_idx[m] : m;
        a_ptr[i] = A + src * lda + ld_chunk * 4;
    }
    const float *b_ptr = B + (int64_t)(n0 + ld_row) * ldb + ld_chunk * 4;

    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ra[4], rb[B_ITERS];
    const int nk = K / BK;

    // prologue: tile 0 -> LDS buffer 0
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(a_ptr[i]);
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr + (int64_t)(32 * i) * ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&As[swz(ld_row + 32 * i, ld_chunk)]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) *reinterpret_cast<f32x4 *>(&Bs[swz(ld_row + 32 * i, ld_chunk)]) = rb[i];
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) {
            const int koff = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(a_ptr[i] + koff);
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i)
                rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr + (int64_t)(32 * i) * ldb + koff);
        }
        const float *Ac = As + cur * BM * BK + (wave * 32) * BK;
        const float *Bc = Bs + cur * BN * BK;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            f32x4 fa[2], fb[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&Ac[swz(i * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&Bc[swz(j * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        // D[n][m]: weights are the MFMA "A" operand so that a lane ends up with 4 consecutive n
                        acc[i][j] = MFMA16(fb[j][s], fa[i][s], acc[i][j]);
        }
        if (more) {
            float *An = As + (cur ^ 1) * BM * BK;
            float *Bn = Bs + (cur ^ 1) * BN * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&An[swz(ld_row + 32 * i, ld_chunk)]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) *reinterpret_cast<f32x4 *>(&Bn[swz(ld_row + 32 * i, ld_chunk)]) = rb[i];
        }
        __syncthreads();
    }

    // epilogue: lane owns C[m = m0 + wave*32 + 16 i + r16][n = n0 + 16 j + 4 q + (0..3)]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wave * 32 + i * 16 + r16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + j * 16 + q * 4;
            f32x4 v = acc[i][j];
            if (EPI == EPI_MASK) {
                const f32x4 h = *reinterpret_cast<const f32x4 *>(mask_src + m * ld_mask + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = h[e] > 0.f ? v[e] : 0.f;
            } else {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[e] + bv[e];
                    if (EPI == EPI_BIAS_RELU) x = x > 0.f ? x : 0.f;
                    if (EPI == EPI_BIAS_TANH) x = tanhf(x);
                    v[e] = x;
                }
            }
            *reinterpret_cast<f32x4 *>(C + m * ldc + n) = v;
        }
    }
}

template <int NB, int EPI>
static int launch_nt_2(hipStream_t st, dim3 grid, const float *A, int64_t lda, const int64_t *row_idx, const float *B,
                       int64_t ldb, const float *bias, const float *mask_src, int64_t ld_mask, float *C, int64_t ldc,
                       int64_t M, int K) {
    if (row_idx)
        hipLaunchKernelGGL((gemm_nt_kernel<NB, EPI, true>), grid, dim3(256), 0, st, A, lda, row_idx, B, ldb, bias,
                           mask_src, ld_mask, C, ldc, M, K);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<NB, EPI, false>), grid, dim3(256), 0, st, A, lda, row_idx, B, ldb, bias,
                           mask_src, ld_mask, C, ldc, M, K);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

template <int NB>
static int launch_nt_1(hipStream_t st, dim3 grid, int epi, const float *A, int64_t lda, const int64_t *row_idx,
                       const float *B, int64_t ldb, const float *bias, const float *mask_src, int64_t ld_mask, float *C,
                       int64_t ldc, int64_t M, int K) {
    switch (epi) {
        case EPI_BIAS: return launch_nt_2<NB, EPI_BIAS>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
        case EPI_BIAS_RELU: return launch_nt_2<NB, EPI_BIAS_RELU>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
        case EPI_BIAS_TANH: return launch_nt_2<NB, EPI_BIAS_TANH>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
        case EPI_MASK: return launch_nt_2<NB, EPI_MASK>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
    }
    set_error("gemm_nt: bad epilogue %d", epi);
    return RLPPO_ERR_ARG;
}

int launch_gemm_nt(hipStream_t st, const float *A, int64_t lda, const int64_t *row_idx, const float *B, int64_t ldb,
                   const float *bias, const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int N,
                   int K, int epi) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(K > 0 && K % BK == 0, "gemm_nt: K=%d must be a positive multiple of %d", K, BK);
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
    dim3 grid((unsigned)cdiv(M, BM), (unsigned)(N / (nb * 16)));
    switch (nb) {
        case 8: return launch_nt_1<8>(st, grid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
        case 6: return launch_nt_1<6>(st, grid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
        case 4: return launch_nt_1<4>(st, grid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
        default: return launch_nt_1<2>(st, grid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K);
    }
}

// ------------------------------------------------------------------------------------------------- gemm_tn
constexpr int TM = 32;        // sample rows per LDS stage
constexpr int TLD = 128 + 16; // LDS row stride (floats): +16 puts rows m and m+1 on opposite bank halves (ds_read_b32)
constexpr int ROWS_PER_WG = 1024;

template <bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const float *__restrict__ dY, int64_t ldy, int ny_valid,
                                                          const float *__restrict__ X, int64_t ldx,
                                                          const int64_t *__restrict__ row_idx, int kx_valid,
                                                          float *__restrict__ dW, float *__restrict__ db, int out,
                                                          int in, int64_t M) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TM * TLD];
    float *Ys = lds;                 // [2][TM][TLD]
    float *Xs = lds + 2 * TM * TLD;  // [2][TM][TLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int64_t mbeg = (int64_t)blockIdx.z * ROWS_PER_WG;
    const int64_t mend = (mbeg + ROWS_PER_WG < M) ? mbeg + ROWS_PER_WG : M;
    const int steps = (int)((mend - mbeg + TM - 1) / TM);

    // staging: chunk = tid & 31 (16 B of a 128-float row), rows (tid >> 5) + 8 i
    const int ld_chunk = tid & 31, ld_row = tid >> 5;
    const bool y_col_ok = (n0 + ld_chunk * 4) < ny_valid;
    const bool x_col_ok = (k0 + ld_chunk * 4) < kx_valid;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;  // thread t < 128 of k-tile 0 accumulates the column sum for db[n0 + t]

    f32x4 ry[4], rx[4];
    auto load_tile = [&](int step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = mbeg + (int64_t)step * TM + ld_row + 8 * i;
            const bool ok = m < mend;
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            ry[i] = (ok && y_col_ok) ? *reinterpret_cast<const f32x4 *>(dY + m * ldy + n0 + ld_chunk * 4) : z;
            if (ok && x_col_ok) {
                const int64_t src = GATHER ? row_idx[m] : m;
                rx[i] = *reinterpret_cast<const f32x4 *>(X + src * ldx + k0 + ld_chunk * 4);
            } else {
                rx[i] = z;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(&Ys[(buf * TM + ld_row + 8 * i) * TLD + ld_chunk * 4]) = ry[i];
            *reinterpret_cast<f32x4 *>(&Xs[(buf * TM + ld_row + 8 * i) * TLD + ld_chunk * 4]) = rx[i];
        }
    };

    if (steps > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        const bool more = (st + 1) < steps;
        if (more) load_tile(st + 1);
        const float *Yc = Ys + cur * TM * TLD;
        const float *Xc = Xs + cur * TM * TLD;
#pragma unroll
        for (int c = 0; c < TM / 16; ++c) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = c * 16 + s * 4 + q;  // the row this lane's k-slot covers in step s
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = Yc[m * TLD + wn * 64 + i * 16 + r16];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = Xc[m * TLD + wk * 64 + j * 16 + r16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(fa[i], fb[j], acc[i][j]);
            }
        }
        if (db != nullptr && blockIdx.y == 0 && tid < 128) {
#pragma unroll 8
            for (int m = 0; m < TM; ++m) bsum += Yc[m * TLD + tid];
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // D[n][k]: lane owns rows n = 16 i + 4 q + e, column k = 16 j + r16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk * 64 + j * 16 + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wn * 64 + i * 16 + q * 4 + e;
                if (n < out && k < in) atomicAdd(dW + (int64_t)n * in + k, acc[i][j][e]);
            }
        }
    if (db != nullptr && blockIdx.y == 0 && tid < 128 && (n0 + tid) < out) atomicAdd(db + n0 + tid, bsum);
}

int launch_gemm_tn(hipStream_t st, const float *dY, int64_t ldy, int ny_valid, const float *X, int64_t ldx,
                   const int64_t *row_idx, int kx_valid, float *dW, float *db, int out, int in, int64_t M) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(ny_valid % 4 == 0 && kx_valid % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ny_valid <= ldy &&
                        kx_valid <= ldx && out <= ny_valid && in <= kx_valid,
                    "gemm_tn: bad shapes ny=%d kx=%d ldy=%ld ldx=%ld out=%d in=%d", ny_valid, kx_valid, (long)ldy,
                    (long)ldx, out, in);
    dim3 grid((unsigned)cdiv(out, 128), (unsigned)cdiv(in, 128), (unsigned)cdiv(M, ROWS_PER_WG));
    if (row_idx)
        hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, dim3(256), 0, st, dY, ldy, ny_valid, X, ldx, row_idx, kx_valid,
                           dW, db, out, in, M);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, dim3(256), 0, st, dY, ldy, ny_valid, X, ldx, row_idx, kx_valid,
                           dW, db, out, in, M);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo
