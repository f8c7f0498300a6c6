// probe.hip -- diagnostic micro-kernels (not on the product path): what does the GEMM inner loop sustain when its
// ingredients are added one at a time?  64 MFMAs (2 x 8 accumulator blocks, the wave tile of every GEMM kernel here)
// per 16-deep k chunk, plus optionally the LDS fragment reads and/or the global fragment loads of that chunk.
#include "../common.hpp"
#include "diag.hpp"

namespace rlppo {
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// mode bit 1: 2 A fragments per chunk from LDS ; bit 2: 8 B fragments per chunk from LDS ; bit 4: 8 B fragments per
// chunk from global memory (256 KB L2-resident matrix, fragment-shaped loads)
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe2_kernel(const float *__restrict__ W, float *__restrict__ out, int chunks) {
    __shared__ __attribute__((aligned(16))) float T[128 * 256];
    f32x4 *T4 = reinterpret_cast<f32x4 *>(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < 128 * 64; i += THREADS) T4[i] = f32x4{0.001f * (i & 255), 0.5f, -0.25f, 0.125f};
    __syncthreads();
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2] = {f32x4{0.1f, 0.2f, 0.3f, 0.4f}, f32x4{0.5f, 0.6f, 0.7f, 0.8f}};
    f32x4 fb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[j] = f32x4{0.01f * j, 0.02f, 0.03f, 0.04f};
    const int row0 = (wave & 3) * 32;
    const float *wp = W + (int64_t)((wave & 1) * 128 + r16) * 256 + q * 4;
    for (int c = 0; c < chunks; ++c) {
        const int kc = c & 15;
        if (MODE & 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(wp + (int64_t)j * 16 * 256 + kc * 16);
        }
        if (MODE & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = row0 + i * 16 + r16;
                fa[i] = T4[r * 64 + ((kc * 4 + q) ^ (r & 15))];
            }
        }
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = j * 16 + r16;
                fb[j] = T4[r * 64 + ((kc * 4 + q) ^ (r & 15))];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = MFMA16(fb[j][s], fa[i][s], acc[i][j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * THREADS + tid] = sum;
}

int launch_probe2(hipStream_t st, int mode, int threads, int blocks, const float *W, float *out, int chunks) {
#define P2(M)                                                                                                 \
    case M:                                                                                                   \
        if (threads == 512)                                                                                   \
            hipLaunchKernelGGL((probe2_kernel<M, 512>), dim3(blocks), dim3(512), 0, st, W, out, chunks);      \
        else                                                                                                  \
            hipLaunchKernelGGL((probe2_kernel<M, 256>), dim3(blocks), dim3(256), 0, st, W, out, chunks);      \
        break;
    switch (mode) {
        P2(0) P2(1) P2(2) P2(3) P2(4) P2(5)
        default:
            set_error("probe2: mode %d", mode);
            return RLPPO_ERR_ARG;
    }
#undef P2
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo

// ------------------------------------------------------------------------------------------------ stamped GEMM
// Copy of gemm_nt_kernel<8, EPI_BIAS_RELU, false, 32> (csrc/gemm.hip) with s_memtime stamps around its phases; lane 0 of
// every wave accumulates cycles per phase and stores them to `stamps[wg][wave][8]`:
//   0 prologue (first loads + LDS write + barrier)   1 issue of next-tile global loads   2 LDS fragment reads + MFMA issue
//   3 wait for the staged loads (vmcnt)              4 LDS writes                        5 barrier
//   6 epilogue (bias loads + stores issued)          7 whole kernel
// Stamped builds are for SHARES, not for absolute time (cdna_hip_programming.md section 7, In-kernel stamps).
namespace rlppo {
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ int pswz(int row, int chunk) { return row * 32 + ((chunk ^ (row & 7)) << 2); }

__global__ __launch_bounds__(256, 2) void gemm_nt_stamped_kernel(const float *__restrict__ A, int64_t lda,
                                                                  const float *__restrict__ B, int64_t ldb,
                                                                  const float *__restrict__ bias, float *__restrict__ C,
                                                                  int64_t ldc, int64_t M, int K,
                                                                  unsigned long long *__restrict__ stamps, int mode) {
    constexpr int NB = 8, BN = 128, BMt = 128, BKT = 32;
    __shared__ __attribute__((aligned(16))) float lds[2 * BMt * BKT + 2 * BN * BKT];
    float *As = lds, *Bs = lds + 2 * BMt * BKT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * BMt;
    const int n0 = blockIdx.y * BN;
    const int ld_chunk = tid & 7, ld_row = tid >> 3;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mode & 16) __builtin_amdgcn_s_setprio(3);
    const unsigned long long tk0 = stamp();
    const float *a_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m0 + ld_row + 32 * i;
        if (m >= M) m = M - 1;
        if (mode & 1) m = (m & 127) + 128 * (blockIdx.x & 7);  // diagnostic: every workgroup reads the same few A rows (L2 hits)
        a_ptr[i] = A + m * lda + ld_chunk * 4;
    }
    const float *b_ptr = B + (int64_t)(n0 + ld_row) * ldb + ld_chunk * 4;
    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[4], rb[4];
    const int nk = K / BKT;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(a_ptr[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr + (int64_t)(32 * i) * ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&As[pswz(ld_row + 32 * i, ld_chunk)]) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&Bs[pswz(ld_row + 32 * i, ld_chunk)]) = rb[i];
    __syncthreads();
    unsigned long long t = stamp();
    acc_t[0] = t - tk0;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) {
            const int koff = (kt + 1) * BKT;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(a_ptr[i] + koff);
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(b_ptr + (int64_t)(32 * i) * ldb + koff);
        }
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t2 = stamp();
        acc_t[1] += t2 - t;
        __builtin_amdgcn_sched_barrier(0);
        const float *Ac = As + cur * BMt * BKT + (wave * 32) * BKT;
        const float *Bc = Bs + cur * BN * BKT;
        if (mode & 16) __builtin_amdgcn_s_setprio(0);
        if (mode & 32) __builtin_amdgcn_s_setprio(3);
        if (!(mode & 4))  // diagnostic: mode 4 skips the fragment reads + MFMAs
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            f32x4 fa[2], fb[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&Ac[pswz(i * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&Bc[pswz(j * 16 + r16, kc * 4 + q)]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = MFMA16(fb[j][s], fa[i][s], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mode & 16) __builtin_amdgcn_s_setprio(3);
        if (mode & 32) __builtin_amdgcn_s_setprio(0);
        unsigned long long t3 = stamp();
        acc_t[2] += t3 - t2;
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t4 = stamp();
            acc_t[3] += t4 - t3;
            __builtin_amdgcn_sched_barrier(0);
            float *An = As + (cur ^ 1) * BMt * BKT, *Bn = Bs + (cur ^ 1) * BN * BKT;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&An[pswz(ld_row + 32 * i, ld_chunk)]) = ra[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&Bn[pswz(ld_row + 32 * i, ld_chunk)]) = rb[i];
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t5 = stamp();
            acc_t[4] += t5 - t4;
            t3 = t5;
        }
        __syncthreads();
        t = stamp();
        acc_t[5] += t - t3;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wave * 32 + i * 16 + r16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + j * 16 + q * 4;
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + n);
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[e] + bv[e];
                v[e] = x > 0.f ? x : 0.f;
            }
            if (!(mode & 2)) *reinterpret_cast<f32x4 *>(C + m * ldc + n) = v;  // diagnostic: mode 2 drops the stores
            else if (v[0] == 12345.678f) C[0] = v[1];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tend = stamp();
    acc_t[6] = tend - t;
    acc_t[7] = tend - tk0;
    if (lane == 0) {
        unsigned long long *o = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = acc_t[k];
    }
}

int launch_gemm_nt_stamped(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                           float *C, int64_t ldc, int64_t M, int N, int K, unsigned long long *stamps, int mode) {
    RLPPO_CHECK_ARG(N % 128 == 0 && K % 32 == 0, "stamped gemm: N %% 128, K %% 32");
    hipLaunchKernelGGL(gemm_nt_stamped_kernel, dim3((unsigned)cdiv(M, 128), N / 128), dim3(256),
                       (mode & 8) ? 48 * 1024 : 0 /* mode 8: one workgroup per CU */, st, A, lda, B, ldb, bias, C,
                       ldc, M, K, stamps, mode);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo

// ------------------------------------------------------------------------------------------------ load-path probe
// How many bytes per clock can one CU pull through its vector-memory path, and does the lane->address pattern of a
// global_load_dwordx4 matter?  Every wave reads 8 KB per iteration with 8 independent loads; the patterns only differ in
// which 16 bytes each lane takes:  0: 8 rows x 128 B (row stride 1 KB; the GEMM A/B staging pattern)   1: 1 KB contiguous
//                                  2: 4 rows x 256 B                                                   3: 16 rows x 64 B
using rlppo::f32x4;
using rlppo::stamp;
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PAT>
__global__ __launch_bounds__(256) void probe_ld_kernel(const char *__restrict__ buf, size_t span, int iters,
                                                       float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    size_t lane_off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (PAT == 0) lane_off[k] = (size_t)(lane >> 3) * 1024 + (lane & 7) * 16 + k * 128;
        if (PAT == 1) lane_off[k] = (size_t)lane * 16 + k * 1024;
        if (PAT == 2) lane_off[k] = (size_t)(lane >> 4) * 1024 + (lane & 15) * 16 + (k & 3) * 256 + (k >> 2) * 4096;
        if (PAT == 3) lane_off[k] = (size_t)(lane >> 2) * 512 + (lane & 3) * 16 + k * 64;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t base = (wave * 8192) & (span - 1);
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4 *>(buf + base + lane_off[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
        base = (base + nwaves * 8192) & (span - 1);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

namespace rlppo {
int launch_probe_ld(hipStream_t st, int pat, int blocks, const void *buf, size_t span, int iters, float *out) {
    if (span < (1u << 16) || (span & (span - 1))) {
        set_error("probe_ld: span must be a power of two >= 64 KiB");
        return RLPPO_ERR_ARG;
    }
    switch (pat) {
        case 0: hipLaunchKernelGGL(probe_ld_kernel<0>, dim3(blocks), dim3(256), 0, st, (const char *)buf, span, iters, out); break;
        case 1: hipLaunchKernelGGL(probe_ld_kernel<1>, dim3(blocks), dim3(256), 0, st, (const char *)buf, span, iters, out); break;
        case 2: hipLaunchKernelGGL(probe_ld_kernel<2>, dim3(blocks), dim3(256), 0, st, (const char *)buf, span, iters, out); break;
        case 3: hipLaunchKernelGGL(probe_ld_kernel<3>, dim3(blocks), dim3(256), 0, st, (const char *)buf, span, iters, out); break;
        default: set_error("probe_ld: pattern %d", pat); return RLPPO_ERR_ARG;
    }
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo

// ------------------------------------------------------------------------------------------------ co-issue probe
// Two roles share every CU (2 workgroups x 256 threads per CU, role = blockIdx.x / (gridDim.x/2)): "streamers" issue a long
// run of independent MFMAs (optionally with LDS fragment reads in between), "loaders" issue batches of 8 global_load_dwordx4
// and record the cycles each batch took to ISSUE (not to return).  Answers: does a wave that streams MFMAs slow down the
// instruction issue of the other wave on its SIMD?   flags: 1 streamer does LDS reads, 2 loader computes 64-bit addresses
// per batch, 4 streamers idle (no MFMA), 8 loaders use priority 3.
__global__ __launch_bounds__(256, 2) void probe_coissue_kernel(const float *__restrict__ buf, int flags, int iters,
                                                               unsigned long long *__restrict__ cycles,
                                                               float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool loader = blockIdx.x >= gridDim.x / 2;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    if (!loader) {
        if (flags & 4) {
            for (int it = 0; it < iters * 8; ++it) __builtin_amdgcn_s_sleep(64);
            return;
        }
        f32x4 acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 fa[2] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}}, fb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
        for (int it = 0; it < iters; ++it) {
            if (flags & 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&lds[((it * 2 + i) * 256 + lane * 4) & 8191]);
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&lds[((it * 8 + j) * 256 + lane * 4) & 8191]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i * 8 + j] = MFMA16(fb[j][s2], fa[i][s2], acc[i * 8 + j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        out[(size_t)blockIdx.x * 256 + tid] = sum;
        return;
    }
    if (flags & 8) __builtin_amdgcn_s_setprio(3);
    const size_t wave = (size_t)(blockIdx.x - gridDim.x / 2) * 4 + (tid >> 6);
    const char *base = reinterpret_cast<const char *>(buf) + ((wave * 8192) & ((1u << 24) - 1));  // 16 MiB window: L2 hits
    const size_t lane_off = (size_t)(lane >> 3) * 1024 + (lane & 7) * 16;
    f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
    unsigned long long issue = 0, total0 = stamp();
    const int batches = iters / 4;
    for (int it = 0; it < batches; ++it) {
        f32x4 v[8];
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = stamp();
        __builtin_amdgcn_sched_barrier(0);
        // flags 16/32/64/128: instead of loads, time a block of 16 VALU instructions of one class
        if (flags & 0xF0) {
            float x0 = sum4[0], x1 = sum4[1];
            unsigned u0 = it, u1 = lane;
            unsigned long long w0 = it;
            f32x2 p0 = {sum4[2], sum4[3]};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (flags & 16) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(u1));
                if (flags & 32) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w0) : "s"((unsigned long long)it));
                if (flags & 64) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p0));
                if (flags & 128) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1));
            }
            __builtin_amdgcn_sched_barrier(0);
            issue += stamp() - t0;
            sum4[0] = x0 + (float)u0 + (float)w0;
            sum4[2] = p0[0] + p0[1];
            __builtin_amdgcn_s_sleep(32);
            continue;
        }
        if (flags & 2) {
            const int64_t k = (int64_t)(it & 7) * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = *reinterpret_cast<const f32x4 *>(base + (int64_t)(it & 1) * 65536 + lane_off + (int64_t)i * 8192 * (1 + (it & 1)) + k);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4 *>(base + lane_off + i * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue += stamp() - t0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) sum4 += v[i];
        __builtin_amdgcn_s_sleep(32);
    }
    const unsigned long long total = stamp() - total0;
    out[(size_t)blockIdx.x * 256 + tid] = sum4[0] + sum4[1] + sum4[2] + sum4[3];
    if (lane == 0) {
        cycles[wave * 2] = issue;
        cycles[wave * 2 + 1] = total;
    }
}

namespace rlppo {
int launch_probe_coissue(hipStream_t st, const float *buf, int flags, int iters, unsigned long long *cycles, float *out) {
    hipLaunchKernelGGL(probe_coissue_kernel, dim3(512), dim3(256), 0, st, buf, flags, iters, cycles, out);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo
