/*
 * rlppo_diag.h -- C ABI of librlppo_diag.so: measurement probes for gfx950 (NOT part of the product library; loaded only by
 * tools/).  They produced the hardware findings quoted in DESIGN.md section 5 (sustained fp32 MFMA rate, vector-memory bytes per
 * clock, the ~400-cycle VALU issue cost beside an MFMA stream, phase shares of a register-staged GEMM).  Same conventions as
 * rlppo.h: device pointers, asynchronous on `stream`, 0 = OK.
 */
#ifndef RLPPO_DIAG_H
#define RLPPO_DIAG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* The export table: the library is built with -fvisibility=hidden, and exactly the functions declared between this push and
 * the matching pop have default visibility (`nm -D` shows them and nothing else of the library's own; tests/test_abi_and_layout.py). */
#pragma GCC visibility push(default)
const char *rlppo_diag_last_error(void);
/* Register-only fp32 MFMA loop: out[blocks*256] floats, clocks[2*blocks] = {shader cycles, 100 MHz ticks} per block. */
int rlppo_dbg_mfma_probe(void *stream, float *out, int32_t blocks, int32_t iters, uint64_t *clocks);
/* GEMM inner-loop probe: 64 MFMAs per chunk + (mode&1) A fragments from LDS, (mode&2) B fragments from LDS, (mode&4) B
 * fragments from a 256x256 fp32 matrix W in global memory.  out: blocks*threads floats. */
int rlppo_dbg_probe2(void *stream, int32_t mode, int32_t threads, int32_t blocks, const float *W, float *out, int32_t chunks);
/* vector-memory path probe: every wave streams 8 KB per iteration with 8 dwordx4 loads; `pattern` picks the lane->address
 * map (0: 8 rows x 128 B, 1: 1 KB contiguous, 2: 4 rows x 256 B, 3: 16 x 64 B); span = power-of-two bytes walked. */
int rlppo_dbg_probe_ld(void *stream, int32_t pattern, int32_t blocks, const void *buf, size_t span, int32_t iters, float *out);
/* co-issue probe: 512 workgroups, the first 256 stream MFMAs, the last 256 issue batches of 8 global loads and record the
 * cycles each batch took to issue -> cycles[wave][2] = {issue, total}.  buf >= 16 MiB, out >= 512*256 floats. */
int rlppo_dbg_probe_coissue(void *stream, const float *buf, int32_t flags, int32_t iters, uint64_t *cycles, float *out);
/* Stamped register-staged forward GEMM (bias+ReLU, N % 128 == 0): stamps[wg][wave][8] cycles per phase.
 * mode: 0 real; 1 every workgroup reads the same 1024 A rows (A from L2); 2 output stores dropped; 3 both */
int rlppo_dbg_gemm_nt_stamped(void *stream, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                              float *C, int64_t ldc, int64_t M, int32_t N, int32_t K, uint64_t *stamps, int32_t mode);
/* Streaming floor of the GAE scan: four fp32 input streams and three output streams of n steps (n % 2048 == 0), 8 steps per thread,
 * an elementwise map instead of the scan: 28 B/step through the memory system and nothing else.  shape 0: a thread owns 8
 * consecutive steps (two float4 at a 32-byte lane stride, the scan's own access shape); shape 1: a thread owns two float4 groups
 * 256 lanes apart, so every wave-instruction moves 1 KiB of contiguous bytes. */
int rlppo_dbg_stream_floor(void *stream, const float *r, const float *d, const float *t, const float *v, float *o0, float *o1,
                           float *o2, int64_t n, int32_t shape);
/* EXPERIMENT (round 4): C[M][256] = relu(A[M][K] . W[256][K]^T + bias) with fp32 A in memory and the products on the
 * bf16 MFMA pipe: A split on the fly into three bf16 pieces (exact), W pre-split into three bf16 planes in stage-major order
 * w_split[K / 32][3][256][32]; `terms` piece products kept per 32-wide K block (6 = all with a piece-index sum <= 2; 4, 3, 1: fewer,
 * to see what each costs / buys).  store = 0 drops the output stores (K loop alone).  M % 256 == 0, K % 32 == 0. */
int rlppo_dbg_gemm_nt_split(void *stream, const float *A, int64_t lda, const void *w_split, const float *bias, float *C, int64_t ldc, int64_t M,
                            int32_t N, int32_t K, int32_t terms, int32_t store);
#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
