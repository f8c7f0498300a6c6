// diag_api.hip -- extern "C" surface of librlppo_diag.so (include/rlppo_diag.h): the measurement probes behind the numbers
// of DESIGN.md section 5 (MFMA ceiling, load-path bandwidth, VALU co-issue cost, phase stamps of a staged GEMM).  Built next
// to librlppo.so, loaded only by tools/; the product library does not contain or need any of it.
#include <stdarg.h>

#include "../../../include/rlppo_diag.h"
#include "diag.hpp"

namespace rlppo {
static thread_local char g_diag_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_diag_err, sizeof(g_diag_err), fmt, ap);
    va_end(ap);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Register-only loop of v_mfma_f32_16x16x4_f32 on 16 independent accumulators (the instruction mix of the GEMM inner
// loops without any memory traffic): measures what the chip sustains on THIS box (clock under load included), so
// that roofline fractions can also be read against an achievable ceiling.
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

// Streaming floor of the GAE scan's traffic: the same four input streams (8 consecutive steps per thread as float4 pairs) and
// three output streams as gae_lookback_kernel, with an elementwise map instead of the scan and no look-ahead window -- what a
// launch of that size and byte count costs when nothing but the memory system is in the way.
__global__ __launch_bounds__(256) void stream_floor_kernel(const float *__restrict__ r, const float *__restrict__ d,
                                                           const float *__restrict__ t, const float *__restrict__ v,
                                                           float *__restrict__ o0, float *__restrict__ o1, float *__restrict__ o2,
                                                           long long n8, int shape) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    // shape 0: a thread owns 8 consecutive steps (two float4 at a 32-byte lane stride: the scan's shape); shape 1: a thread owns two
    // float4 groups 256 lanes apart, so that every wave-instruction moves 1 KiB of contiguous bytes
    // shapes 2 / 3 [r4]: lane-contiguous loads with 8-consecutive stores / the reverse (timing only: which half of the difference
    // between shapes 0 and 1 belongs to the loads)
    const bool lc = shape == 1 || shape == 2, sc = shape == 1 || shape == 3;
    const long long base = lc ? (long long)blockIdx.x * 512 + threadIdx.x : 2 * i;
    const long long step = lc ? 256 : 1;
    const long long sbase = sc ? (long long)blockIdx.x * 512 + threadIdx.x : 2 * i;
    const long long sstep = sc ? 256 : 1;
    const float4 *r4 = (const float4 *)r + base, *d4 = (const float4 *)d + base, *t4 = (const float4 *)t + base,
                 *v4 = (const float4 *)v + base;
    float4 a[2], b[2], c[2], e[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { a[k] = r4[k * step]; b[k] = d4[k * step]; c[k] = t4[k * step]; e[k] = v4[k * step]; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float4 x, y, z;
        x.x = a[k].x + 0.99f * e[k].x * (1.f - b[k].x); x.y = a[k].y + 0.99f * e[k].y * (1.f - b[k].y);
        x.z = a[k].z + 0.99f * e[k].z * (1.f - b[k].z); x.w = a[k].w + 0.99f * e[k].w * (1.f - b[k].w);
        y.x = x.x * (1.f - c[k].x); y.y = x.y * (1.f - c[k].y); y.z = x.z * (1.f - c[k].z); y.w = x.w * (1.f - c[k].w);
        z.x = y.x + e[k].x; z.y = y.y + e[k].y; z.z = y.z + e[k].z; z.w = y.w + e[k].w;
        ((float4 *)o0)[sbase + k * sstep] = x; ((float4 *)o1)[sbase + k * sstep] = y; ((float4 *)o2)[sbase + k * sstep] = z;
    }
}
int launch_stream_floor(hipStream_t st, const float *r, const float *d, const float *t, const float *v, float *o0, float *o1,
                        float *o2, long long n, int shape) {
    const long long n8 = n / 8;
    hipLaunchKernelGGL(stream_floor_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, r, d, t, v, o0, o1, o2, n8, shape);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
}  // namespace rlppo

using namespace rlppo;

extern "C" {
const char *rlppo_diag_last_error(void) { return g_diag_err; }
int rlppo_dbg_gemm_nt_stamped(void *stream, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                              float *C, int64_t ldc, int64_t M, int32_t N, int32_t K, uint64_t *stamps, int32_t mode) {
    return launch_gemm_nt_stamped((hipStream_t)stream, A, lda, B, ldb, bias, C, ldc, M, N, K, (unsigned long long *)stamps, mode);
}
int rlppo_dbg_probe_ld(void *stream, int32_t pattern, int32_t blocks, const void *buf, size_t span, int32_t iters, float *out) {
    return launch_probe_ld((hipStream_t)stream, pattern, blocks, buf, span, iters, out);
}
int rlppo_dbg_probe_coissue(void *stream, const float *buf, int32_t flags, int32_t iters, uint64_t *cycles, float *out) {
    return launch_probe_coissue((hipStream_t)stream, buf, flags, iters, (unsigned long long *)cycles, out);
}
int rlppo_dbg_probe2(void *stream, int32_t mode, int32_t threads, int32_t blocks, const float *W, float *out, int32_t chunks) {
    return launch_probe2((hipStream_t)stream, mode, threads, blocks, W, out, chunks);
}
int rlppo_dbg_mfma_probe(void *stream, float *out, int32_t blocks, int32_t iters, uint64_t *clocks) {
    return launch_mfma_probe((hipStream_t)stream, out, blocks, iters, (unsigned long long *)clocks);
}
int rlppo_dbg_stream_floor(void *stream, const float *r, const float *d, const float *t, const float *v, float *o0, float *o1,
                           float *o2, int64_t n, int32_t shape) {
    if (n <= 0 || n % 2048) return 1001;
    return launch_stream_floor((hipStream_t)stream, r, d, t, v, o0, o1, o2, (long long)n, shape);
}
int rlppo_dbg_gemm_nt_split(void *stream, const float *A, int64_t lda, const void *w_split, const float *bias, float *C, int64_t ldc, int64_t M,
                            int32_t N, int32_t K, int32_t terms, int32_t store) {
    return launch_gemm_nt_split((hipStream_t)stream, A, lda, reinterpret_cast<const unsigned short *>(w_split), bias, C, ldc, M, N, K, terms, store);
}
}
