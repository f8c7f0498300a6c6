// diag.hpp -- launchers of librlppo_diag.so (measurement probes; nothing here is on the product path).
#pragma once
#include "../common.hpp"

namespace rlppo {
int launch_gemm_nt_stamped(hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                           float *C, int64_t ldc, int64_t M, int N, int K, unsigned long long *stamps, int mode);
int launch_probe_ld(hipStream_t st, int pat, int blocks, const void *buf, size_t span, int iters, float *out);
int launch_probe_coissue(hipStream_t st, const float *buf, int flags, int iters, unsigned long long *cycles, float *out);
int launch_probe2(hipStream_t st, int mode, int threads, int blocks, const float *W, float *out, int chunks);
int launch_mfma_probe(hipStream_t st, float *out, int blocks, int iters, unsigned long long *clocks);
int launch_stream_floor(hipStream_t st, const float *r, const float *d, const float *t, const float *v, float *o0, float *o1,
                        float *o2, long long n, int shape);
int launch_gemm_nt_split(hipStream_t st, const float *A, int64_t lda, const unsigned short *Ws, const float *bias, float *C, int64_t ldc,
                         int64_t M, int N, int K, int terms, int store);
}  // namespace rlppo
