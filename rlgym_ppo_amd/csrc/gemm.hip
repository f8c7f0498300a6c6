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

// LDS tile of R rows x BKT floats in 16-byte chunks, XOR-swizzled so that a ds_read_b128 fragment read (16 rows x
// 4 chunk columns per instruction, issued to four 16-lane groups) touches 16 distinct 16-byte slots of the 256-byte
// bank row -> conflict free (checked with SQ_LDS_BANK_CONFLICT = 0, profiles/r01_pmc_sq_v2.csv).
//   BKT = 32: 8 chunks per row,  chunk ^ (row & 7)
//   BKT = 16: 4 chunks per row,  chunk ^ ((-(row >> 2)) & 3)
template <int BKT>
__device__ __forceinline__ int swz(int row, int chunk) {
    if (BKT == 32) return row * 32 + ((chunk ^ (row & 7)) << 2);
    return row * 16 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 2);
}

template <int NB, int EPI, bool GATHER, int BKT>
__global__ __launch_bounds__(256, BKT == 16 ? 3 : 2) void gemm_nt_kernel(const float *__restrict__ A, int64_t lda,
                                                          const int64_t *__restrict__ row_idx,
                                                          const float *__restrict__ B, int64_t ldb,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ mask_src, int64_t ld_mask,
                                                          float *__restrict__ C, int64_t ldc, int64_t M, int K,
                                                          int stagger) {
    constexpr int BN = NB * 16;
    constexpr int CPR = BKT / 4;          // 16-byte chunks per staged row
    constexpr int RPP = 256 / CPR;        // rows staged per pass of the 256 threads
    constexpr int A_ITERS = BM / RPP;
    constexpr int B_ITERS = BN / RPP;
    static_assert(BN % RPP == 0, "column tile must be a whole number of staging passes");
    __shared__ __attribute__((aligned(16))) float lds[2 * BM * BKT + 2 * BN * BKT];
    float *As = lds;
    float *Bs = lds + 2 * BM * BKT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // staging assignment: chunk = tid % CPR (16 B), rows tid / CPR + RPP * i
    const int ld_chunk = tid % CPR, ld_row = tid / CPR;
    const float *a_ptr[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        int64_t m = m0 + ld_row + RPP * i;
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

    f32x4 ra[A_ITERS], rb[B_ITERS];
    const int nk = K / BKT;

    // Two workgroups share a CU and run identical code, so they reach their barrier / staging phases together and the
    // MFMA pipe idles in those phases.  Delaying every second wave of workgroups (blocks 256..511, 768..1023, ...: the
    // ones that land as the SECOND workgroup of a CU under the observed round-robin dispatch; speed only, never
    // correctness) by about half a K step puts the pair out of phase for its whole life.
    if (stagger > 0 && (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1))
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(8);  // 8 x 64 cycles per unit
    // prologue: tile 0 -> LDS buffer 0
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(a_ptr[i]);
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr + (int64_t)(RPP * i) * ldb);
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) *reinterpret_cast<f32x4 *>(&As[swz<BKT>(ld_row + RPP * i, ld_chunk)]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) *reinterpret_cast<f32x4 *>(&Bs[swz<BKT>(ld_row + RPP * i, ld_chunk)]) = rb[i];
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) {
            const int koff = (kt + 1) * BKT;
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(a_ptr[i] + koff);
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i)
                rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr + (int64_t)(RPP * i) * ldb + koff);
        }
        const float *Ac = As + cur * BM * BKT + (wave * 32) * BKT;
        const float *Bc = Bs + cur * BN * BKT;
#pragma unroll
        for (int kc = 0; kc < BKT / 16; ++kc) {
            f32x4 fa[2], fb[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&Ac[swz<BKT>(i * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&Bc[swz<BKT>(j * 16 + r16, kc * 4 + q)]);
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
            float *An = As + (cur ^ 1) * BM * BKT;
            float *Bn = Bs + (cur ^ 1) * BN * BKT;
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) *reinterpret_cast<f32x4 *>(&An[swz<BKT>(ld_row + RPP * i, ld_chunk)]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) *reinterpret_cast<f32x4 *>(&Bn[swz<BKT>(ld_row + RPP * i, ld_chunk)]) = rb[i];
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

static int g_nt_bk = 32;  // tuning: rlppo_dbg_set(5, 16 | 32): K step of the staged kernel
static int g_nt_stagger = 0;  // tuning: rlppo_dbg_set(7, units of 512 cycles)
void set_nt_bk(int v) { g_nt_bk = v; }
void set_nt_stagger(int v) { g_nt_stagger = v; }

template <int NB, int EPI>
static int launch_nt_2(hipStream_t st, dim3 grid, const float *A, int64_t lda, const int64_t *row_idx, const float *B,
                       int64_t ldb, const float *bias, const float *mask_src, int64_t ld_mask, float *C, int64_t ldc,
                       int64_t M, int K) {
    constexpr bool can16 = (NB * 16) % 64 == 0;  // BK=16 stages 64 rows per pass
    if (can16 && g_nt_bk == 16) {
        if (row_idx)
            hipLaunchKernelGGL((gemm_nt_kernel<NB, EPI, true, can16 ? 16 : 32>), grid, dim3(256), 0, st, A, lda, row_idx, B,
                               ldb, bias, mask_src, ld_mask, C, ldc, M, K, g_nt_stagger);
        else
            hipLaunchKernelGGL((gemm_nt_kernel<NB, EPI, false, can16 ? 16 : 32>), grid, dim3(256), 0, st, A, lda, row_idx,
                               B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, g_nt_stagger);
    } else if (row_idx)
        hipLaunchKernelGGL((gemm_nt_kernel<NB, EPI, true, 32>), grid, dim3(256), 0, st, A, lda, row_idx, B, ldb, bias,
                           mask_src, ld_mask, C, ldc, M, K, g_nt_stagger);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<NB, EPI, false, 32>), grid, dim3(256), 0, st, A, lda, row_idx, B, ldb, bias,
                           mask_src, ld_mask, C, ldc, M, K, g_nt_stagger);
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

// --------------------------------------------------------------------------------- gemm_nt, weights-stationary
// Second form of the same product for K <= 256 (every layer of the 256x3 nets): the whole B column tile
// ([16*NB rows][K] fp32, <= 128 KB) is loaded into LDS ONCE per workgroup and stays there while the workgroup walks
// over row tiles; every wave streams its own 32 A rows straight from global memory into MFMA fragments (one dwordx4
// per 16x16 block and 16-deep k chunk, 4 chunks in flight), so the main loop has NO barrier and the 8 waves of a
// workgroup drift apart instead of stalling together (the lock-step stalls of the staged kernel above cost ~35 % of
// the MFMA pipe, profiles/r01_pmc_sq_v2.csv).  Row tiles are processed as one flattened (tile, k-chunk) pipeline so
// the loads of the next tile are in flight while the current tile's epilogue stores drain.
constexpr int WS_WAVES = 8;
constexpr int WS_ROWS = WS_WAVES * 32;  // 256 rows per row tile
constexpr int WS_DEPTH = 4;             // k chunks (16 floats) of A kept in flight per wave

// Off by default: measured inside the whole update (tools/ab_update.py) the staged kernel is 5 % faster, because with
// two kernels in flight (policy / critic chains) the staged form's 64 KB workgroups of both kernels share a CU, while
// the stationary form's 128 KB workgroups cannot.  Kept selectable: rlppo_dbg_set(3, 1).
static int g_nt_ws = 0;
void set_nt_ws(int v) { g_nt_ws = v; }

template <int NB, int EPI, bool GATHER>
__global__ __launch_bounds__(512) void gemm_nt_ws_kernel(const float *__restrict__ A, int64_t lda,
                                                          const int64_t *__restrict__ row_idx,
                                                          const float *__restrict__ B, int64_t ldb,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ mask_src, int64_t ld_mask,
                                                          float *__restrict__ C, int64_t ldc, int64_t M, int K,
                                                          int n_row_tiles) {
    constexpr int BN = NB * 16;
    __shared__ __attribute__((aligned(16))) float Bs[BN * 256];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.y * BN;
    const int kch = K >> 2;                       // 16-byte chunks per B row
    const int swm = (kch & 15) == 0 ? 15 : 7;     // XOR swizzle mask (K is a multiple of 32, so kch % 8 == 0)

    // ---- stationary operand: B[n0 .. n0+BN)[0..K) -> LDS, chunk c of row r stored at chunk c ^ (r & swm)
    // (indexing Bs in float4 units tells the compiler the accesses are 16-byte aligned: ds_read_b128 / ds_write_b128)
    f32x4 *Bs4 = reinterpret_cast<f32x4 *>(Bs);
    for (int base = tid; base < BN * kch; base += 512 * 8) {  // 8 independent loads in flight per thread
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = base + u * 512;
            if (id < BN * kch) {
                const int r = id / kch, c = id - r * kch;
                v[u] = *reinterpret_cast<const f32x4 *>(B + (int64_t)(n0 + r) * ldb + c * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = base + u * 512;
            if (id < BN * kch) {
                const int r = id / kch, c = id - r * kch;
                Bs4[r * kch + (c ^ (r & swm))] = v[u];
            }
        }
    }
    __syncthreads();

    const int nkc = K >> 4;                       // 16-float k chunks per tile
    const int my_tiles = (n_row_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nkc;             // flattened (tile, chunk) steps of this workgroup

    // row pointers of the tile currently being LOADED (two 16-row blocks per wave)
    auto tile_row = [&](int t_local, int i) -> int64_t {
        int64_t m = ((int64_t)blockIdx.x + (int64_t)t_local * gridDim.x) * WS_ROWS + wave * 32 + i * 16 + r16;
        return m < M ? m : M - 1;  // clamp: rows past M are computed but never stored
    };
    const float *pa[2];
    int64_t nxt_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = tile_row(0, i);
        pa[i] = A + (GATHER ? row_idx[m] : m) * lda + q * 4;
        nxt_src[i] = GATHER ? row_idx[tile_row(1 < my_tiles ? 1 : 0, i)] : 0;
    }
    int ld_tile = 0, ld_kc = 0;  // position of the next load in the flattened sequence

    f32x4 ring[WS_DEPTH][2];
    auto issue = [&](f32x4 (&dst)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) dst[i] = *reinterpret_cast<const f32x4 *>(pa[i] + ld_kc * 16);
        if (++ld_kc == nkc) {  // advance to the next tile of this workgroup
            ld_kc = 0;
            ++ld_tile;
            const int t = ld_tile < my_tiles ? ld_tile : my_tiles - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t m = tile_row(t, i);
                pa[i] = A + (GATHER ? nxt_src[i] : m) * lda + q * 4;
                if (GATHER) nxt_src[i] = row_idx[tile_row(t + 1 < my_tiles ? t + 1 : t, i)];
            }
        }
    };
#pragma unroll
    for (int d = 0; d < WS_DEPTH; ++d) issue(ring[d]);  // loads past the end re-read the last tile (harmless)

    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int cs_tile = 0, cs_kc = 0;  // position of the chunk being consumed
    for (int step = 0; step < total; step += WS_DEPTH) {
#pragma unroll
        for (int u = 0; u < WS_DEPTH; ++u) {
            if (step + u < total) {  // wave-uniform
                f32x4 fa[2] = {ring[u][0], ring[u][1]};
                issue(ring[u]);  // refill this ring slot with the chunk WS_DEPTH steps ahead
                f32x4 fb[NB];
                const int c = cs_kc * 4 + q;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int r = j * 16 + r16;
                    fb[j] = Bs4[r * kch + (c ^ (r & swm))];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) acc[i][j] = MFMA16(fb[j][s], fa[i][s], acc[i][j]);
                if (++cs_kc == nkc) {  // tile finished: epilogue, then fresh accumulators
                    cs_kc = 0;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int64_t m = ((int64_t)blockIdx.x + (int64_t)cs_tile * gridDim.x) * WS_ROWS + wave * 32 + i * 16 + r16;
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const int n = n0 + j * 16 + q * 4;
                            f32x4 v = acc[i][j];
                            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (m < M) {
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
                    ++cs_tile;
                }
            }
        }
    }
}

template <int NB, int EPI>
static int launch_ws_2(hipStream_t st, dim3 grid, const float *A, int64_t lda, const int64_t *row_idx, const float *B,
                       int64_t ldb, const float *bias, const float *mask_src, int64_t ld_mask, float *C, int64_t ldc,
                       int64_t M, int K, int n_row_tiles) {
    if (row_idx)
        hipLaunchKernelGGL((gemm_nt_ws_kernel<NB, EPI, true>), grid, dim3(512), 0, st, A, lda, row_idx, B, ldb, bias,
                           mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
    else
        hipLaunchKernelGGL((gemm_nt_ws_kernel<NB, EPI, false>), grid, dim3(512), 0, st, A, lda, row_idx, B, ldb, bias,
                           mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

template <int NB>
static int launch_ws_1(hipStream_t st, dim3 grid, int epi, const float *A, int64_t lda, const int64_t *row_idx,
                       const float *B, int64_t ldb, const float *bias, const float *mask_src, int64_t ld_mask, float *C,
                       int64_t ldc, int64_t M, int K, int n_row_tiles) {
    switch (epi) {
        case EPI_BIAS: return launch_ws_2<NB, EPI_BIAS>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
        case EPI_BIAS_RELU: return launch_ws_2<NB, EPI_BIAS_RELU>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
        case EPI_BIAS_TANH: return launch_ws_2<NB, EPI_BIAS_TANH>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
        case EPI_MASK: return launch_ws_2<NB, EPI_MASK>(st, grid, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
    }
    set_error("gemm_nt: bad epilogue %d", epi);
    return RLPPO_ERR_ARG;
}

int launch_gemm_nt(hipStream_t st, const float *A, int64_t lda, const int64_t *row_idx, const float *B, int64_t ldb,
                   const float *bias, const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int N,
                   int K, int epi, int bf16_operands) {
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
    if (bf16_operands && !row_idx && epi != EPI_MASK && 129 * lda * 4 < ((int64_t)1 << 31) && 129 * ldc * 4 < ((int64_t)1 << 31))
        return launch_gemm_nt_bf16(st, A, lda, B, ldb, bias, C, ldc, M, N, nb, K, epi);  // inference-only forward (gemm_sa.hip)
    if (!row_idx && !g_nt_ws) {
        // default for ungathered operands: the scalar-addressed kernel (gemm_sa.hip); -1 = not applicable
        const int rc = launch_gemm_nt_sa(st, A, lda, B, ldb, bias, mask_src, ld_mask, C, ldc, M, N, nb, K, epi);
        if (rc != -1) return rc;
    }
    if (g_nt_ws && K <= 256 && M >= 4 * WS_ROWS) {
        // one workgroup per CU: grid.x * (column tiles) ~ number of CUs, each workgroup walks several row tiles
        const int n_row_tiles = (int)cdiv(M, WS_ROWS);
        const int col_tiles = N / (nb * 16);
        int gx = 256 / col_tiles;
        if (gx > n_row_tiles) gx = n_row_tiles;
        dim3 wgrid((unsigned)gx, (unsigned)col_tiles);
        switch (nb) {
            case 8: return launch_ws_1<8>(st, wgrid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
            case 6: return launch_ws_1<6>(st, wgrid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
            case 4: return launch_ws_1<4>(st, wgrid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
            default: return launch_ws_1<2>(st, wgrid, epi, A, lda, row_idx, B, ldb, bias, mask_src, ld_mask, C, ldc, M, K, n_row_tiles);
        }
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
static int g_tn_rows_override = 0;  // tuning: rlppo_dbg_set(2, rows)
void set_tn_rows(int r) { g_tn_rows_override = r; }
// tuning: rlppo_dbg_set(12, rows): split of products with >= 4 output tiles.  With the LDS-DMA kernel and the two chains
// overlapping (bench.py, M samples/s): 512 rows 49.2, 640 50.4, 768 50.7-50.9, 896 50.3, 1024 49.8
static int g_tn_rows_big = 768;
void set_tn_rows_big(int r) { g_tn_rows_big = r; }
constexpr int TM = 32;        // sample rows per LDS stage
constexpr int TLD = 128 + 16; // LDS row stride (floats): +16 puts rows m and m+1 on opposite bank halves (ds_read_b32)

template <bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const float *__restrict__ dY, int64_t ldy, int ny_valid,
                                                          const float *__restrict__ X, int64_t ldx,
                                                          const int64_t *__restrict__ row_idx, int kx_valid,
                                                          float *__restrict__ dW, float *__restrict__ db, int out,
                                                          int in, int64_t M, int rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TM * TLD];
    float *Ys = lds;                 // [2][TM][TLD]
    float *Xs = lds + 2 * TM * TLD;  // [2][TM][TLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int64_t mbeg = (int64_t)blockIdx.z * rows_per_wg;
    const int64_t mend = (mbeg + rows_per_wg < M) ? mbeg + rows_per_wg : M;
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
    // db: every thread keeps the column sums of the dY rows IT stages (4 columns), straight from the staging registers
    f32x4 bs4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_db = db != nullptr && blockIdx.y == 0;

    f32x4 ry[4], rx[4];
    // gathered X rows: the row indices of stage st+2 are fetched while the data of stage st+1 is in flight, so the
    // dependent index -> address -> data chain costs one memory latency per stage instead of two
    int64_t src_next[4];
    auto load_idx = [&](int step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = mbeg + (int64_t)step * TM + ld_row + 8 * i;
            src_next[i] = (GATHER && m < mend) ? row_idx[m] : 0;
        }
    };
    auto load_tile = [&](int step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = mbeg + (int64_t)step * TM + ld_row + 8 * i;
            const bool ok = m < mend;
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            ry[i] = (ok && y_col_ok) ? *reinterpret_cast<const f32x4 *>(dY + m * ldy + n0 + ld_chunk * 4) : z;
            const int64_t src = GATHER ? src_next[i] : m;
            rx[i] = (ok && x_col_ok) ? *reinterpret_cast<const f32x4 *>(X + src * ldx + k0 + ld_chunk * 4) : z;
        }
        if (GATHER) load_idx(step + 1);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(&Ys[(buf * TM + ld_row + 8 * i) * TLD + ld_chunk * 4]) = ry[i];
            *reinterpret_cast<f32x4 *>(&Xs[(buf * TM + ld_row + 8 * i) * TLD + ld_chunk * 4]) = rx[i];
            if (want_db) bs4 += ry[i];
        }
    };

    if (steps > 0) {
        load_idx(0);
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
    if (want_db) {  // 8 threads (tid >> 5) hold partial sums of the same 4 columns: fold them through LDS
        float *red = lds;  // [8][128]; the staging buffers are dead (the loop ended with a barrier)
        *reinterpret_cast<f32x4 *>(&red[ld_row * 128 + ld_chunk * 4]) = bs4;
        __syncthreads();
        if (tid < 128 && (n0 + tid) < out) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) sum += red[r * 128 + tid];
            atomicAdd(db + n0 + tid, sum);
        }
    }
}

int launch_gemm_tn(hipStream_t st, const float *dY, int64_t ldy, int ny_valid, const float *X, int64_t ldx,
                   const int64_t *row_idx, int kx_valid, float *dW, float *db, int out, int in, int64_t M, float *ws,
                   size_t ws_floats) {
    if (M <= 0) return 0;
    RLPPO_CHECK_ARG(ny_valid % 4 == 0 && kx_valid % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ny_valid <= ldy &&
                        kx_valid <= ldx && out <= ny_valid && in <= kx_valid,
                    "gemm_tn: bad shapes ny=%d kx=%d ldy=%ld ldx=%ld out=%d in=%d", ny_valid, kx_valid, (long)ldy,
                    (long)ldx, out, in);
    if (!row_idx && ws) {  // default with a workspace: partial tiles + reduction instead of atomics (gemm_sa.hip)
        const int rc = launch_gemm_tn_partial(st, dY, ldy, ny_valid, X, ldx, kx_valid, dW, db, out, in, M, ws, ws_floats);
        if (rc != -1) return rc;
    }
    // Split over the sample axis.  Measured on MI355X at M = 65,536 (tools/sweep_tn.py, us per launch):
    //   rows/WG      128    256    512   1024   2048
    //   256x256     157.5  123.9  103.0  104.6  197.5      few splits -> idle CUs; many splits -> fp32 atomic traffic
    //   256x107     159.2  110.4  103.6  149.6  265.2
    //   90x256       83.6   62.8   62.7   98.7  187.6
    //   1x256        48.6   44.6   52.7   95.7  188.3
    const int64_t tiles = cdiv(out, 128) * cdiv(in, 128);
    int64_t rows = tiles >= 2 ? 512 : 256;
    if (tiles >= 4 && g_tn_rows_big > 0) rows = g_tn_rows_big;
    if (M < 64 * rows) rows = round_up(cdiv(M, 64) > 32 ? cdiv(M, 64) : 32, TM);  // small M: still use the chip
    if (g_tn_rows_override > 0) rows = g_tn_rows_override;
    const int rows_per_wg = (int)rows;
    dim3 grid((unsigned)cdiv(out, 128), (unsigned)cdiv(in, 128), (unsigned)cdiv(M, rows_per_wg));
    if (!row_idx) {  // default for ungathered X: the scalar-addressed kernel (gemm_sa.hip); -1 = not applicable
        const int rc = launch_gemm_tn_sa(st, grid, dY, ldy, ny_valid, X, ldx, kx_valid, dW, db, out, in, M, rows_per_wg);
        if (rc != -1) return rc;
    }
    if (row_idx)
        hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, dim3(256), 0, st, dY, ldy, ny_valid, X, ldx, row_idx, kx_valid,
                           dW, db, out, in, M, rows_per_wg);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, dim3(256), 0, st, dY, ldy, ny_valid, X, ldx, row_idx, kx_valid,
                           dW, db, out, in, M, rows_per_wg);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo

// ------------------------------------------------------------------------------------------ MFMA ceiling probe
// Register-only loop of v_mfma_f32_16x16x4_f32 on 16 independent accumulators (the instruction mix of the GEMM inner
// loops without any memory traffic): measures what the chip sustains on THIS box (clock under load included), so
// that roofline fractions can also be read against an achievable ceiling.  Diagnostic entry point only.
namespace rlppo {
__global__ __launch_bounds__(256) void mfma_probe_kernel(float *out, int iters, unsigned long long *clocks) {
    f32x4 acc[16];
    float a[4], b[4];
    const float seed = (float)(threadIdx.x % 37) * 0.03125f - 0.5f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{seed, -seed, 0.5f * seed, 0.25f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s] = seed * (float)(s + 1) * 0.37f + 0.01f;
        b[s] = 0.91f - seed * (float)(s + 1) * 0.11f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = MFMA16(a[s], b[(s + i) & 3], acc[i]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && clocks) {
        clocks[2 * blockIdx.x] = t1 - t0;      // shader cycles
        clocks[2 * blockIdx.x + 1] = r1 - r0;  // 100 MHz ticks
    }
}
int launch_mfma_probe(hipStream_t st, float *out, int blocks, int iters, unsigned long long *clocks) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, out, iters, clocks);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo
