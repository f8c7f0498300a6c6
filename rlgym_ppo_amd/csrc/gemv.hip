// gemv.hip -- the critic's one-output head (value_estimator.py: Linear(hidden, 1)) as three HBM-bound streaming kernels.
//
// Through the GEMM kernels the head is padded to 32 outputs: the forward, the masked dX and the dW launch then spend
// 14 + 29 + 40 us per 65,536-row minibatch on products that are 1/32 real work.  As matrix-vector operations they only
// have to stream the last hidden activation h[M][K] once each:
//   forward : v[m]      = b + sum_k h[m][k] w[k]                      (read  M K 4 bytes)
//   dX      : dx[m][k]  = dv[m] w[k] [h[m][k] > 0]                    (read + write M K 4 bytes; bit-identical to the GEMM:
//                                                                      the other 31 padded terms of its sum are zeros)
//   dW, db  : dw[k]    += sum_m dv[m] h[m][k],  db += sum_m dv[m]     (read  M K 4 bytes, one atomic per column and block)
// v / dv live in column 0 of the critic's padded output buffer (row stride ldv), exactly where the GEMM path keeps them.
#include "common.hpp"

namespace rlppo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
}  // namespace

// One wave per row, FOUR rows in flight per wave (four independent 16-byte loads per lane and 256-float column block before
// anything is consumed), and the four row sums are reduced together: a butterfly that halves the number of live sums at each
// of the first two exchange steps (lane ^ 32 keeps two of the four, lane ^ 16 one of the two) and then finishes the single
// remaining sum, 7 cross-lane exchanges instead of 24.  Lanes 0-15 end up with row 0's sum, 16-31 row 1's, 32-47 row 2's,
// 48-63 row 3's.  The summation order differs from a plain per-row tree only in its grouping (fp32 sums, 2e-7 relative).
__global__ __launch_bounds__(256) void gemv_fwd_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ w,
                                                       const float *__restrict__ b, float *__restrict__ y, int64_t ldy,
                                                       int64_t n, int kp, int pout) {
    const int lane = threadIdx.x & 63;
    const float bias = b[0];
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; row < n; row += (int64_t)gridDim.x * 16) {
        const float *xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xr[u] = x + (row + u < n ? row + u : row) * ldx;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 4; c < kp; c += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c);
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xr[u] + c));
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += v[u][0] * wv[0] + v[u][1] * wv[1] + v[u][2] * wv[2] + v[u][3] * wv[3];
        }
        // step 1 (lane ^ 32): the lower half keeps rows 0, 1 and receives the upper half's partial sums of them, and vice versa
        const bool hi = lane >= 32;
        float s0 = (hi ? a[2] : a[0]) + __shfl_xor(hi ? a[0] : a[2], 32);
        float s1 = (hi ? a[3] : a[1]) + __shfl_xor(hi ? a[1] : a[3], 32);
        // step 2 (lane ^ 16): within each half, the lower quarter keeps the first of its two rows
        const bool q1 = (lane & 16) != 0;
        float s = (q1 ? s1 : s0) + __shfl_xor(q1 ? s0 : s1, 16);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const int r = lane >> 4;  // the row (of the four) whose sum this 16-lane group holds
        if (row + r < n && (lane & 15) * 4 < pout)  // the padded output columns are part of the layout contract: zeros
            *reinterpret_cast<f32x4 *>(y + (row + r) * ldy + (lane & 15) * 4) = f32x4{(lane & 15) == 0 ? s + bias : 0.f, 0.f, 0.f, 0.f};
    }
}

// 16 bytes per thread
__global__ __launch_bounds__(256) void gemv_dx_kernel(const float *__restrict__ dy, int64_t ldy, const float *__restrict__ w,
                                                      const float *__restrict__ mask, int64_t ldm, float *__restrict__ dx,
                                                      int64_t ldc, int cpr, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / cpr;
    const int c = (int)(t - row * cpr) * 4;
    if (row >= n) return;
    const float d = dy[row * ldy];
    const f32x4 h = *reinterpret_cast<const f32x4 *>(mask + row * ldm + c);
    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h[e] > 0.f ? d * wv[e] : 0.f;
    *reinterpret_cast<f32x4 *>(dx + row * ldc + c) = o;
}

// The same product with the ReLU mask taken from the bitmask the hidden layer's forward GEMM wrote (csrc/gemm.hip,
// relu_bits): no read of the activation at all.  A block is one 128 x 128 tile of dx and a thread is the lane that owned the
// same 64 elements in the forward epilogue: wave w, lane (q, r16) -> rows m0 + 32 w + 16 i + r16, columns n0 + 16 j + 4 q + e,
// bit (8 i + j) * 4 + e of the word at bits[tile * 256 + tid] (512 contiguous bytes per wave).
__global__ __launch_bounds__(256) void gemv_dx_bits_kernel(const float *__restrict__ dy, int64_t ldy, const float *__restrict__ w,
                                                           const unsigned long long *__restrict__ bits, float *__restrict__ dx,
                                                           int64_t ldc, int64_t n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    const unsigned long long word = bits[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 256 + tid];
    f32x4 wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = *reinterpret_cast<const f32x4 *>(w + n0 + 16 * j + 4 * q);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t row = m0 + wave * 32 + 16 * i + r16;
        if (row >= n) continue;
        const float d = dy[row * ldy];
        const unsigned half = (unsigned)(word >> (32 * i));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((half >> (j * 4 + e)) & 1u) ? d * wv[j][e] : 0.f;
            *reinterpret_cast<f32x4 *>(dx + row * ldc + n0 + 16 * j + 4 * q) = o;
        }
    }
}

// block = rows [r0, r0 + rows_per_block); thread = (row lane, 16-byte column chunk); cpr chunks per row, 256 / cpr row lanes
__global__ __launch_bounds__(256) void gemv_dw_kernel(const float *__restrict__ dy, int64_t ldy, const float *__restrict__ x,
                                                      int64_t ldx, float *__restrict__ dw, float *__restrict__ db, int in,
                                                      int cpr, int64_t n, int rows_per_block, float *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    __shared__ float dsum_s[4];
    const int chunk = threadIdx.x % cpr, rlane = threadIdx.x / cpr, RL = 256 / cpr;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;
    int64_t m = r0 + rlane;
    for (; m + 7 * RL < r1; m += 8 * RL) {  // eight independent rows in flight per thread (the activation is read once: nt)
        float d[8];
        f32x4 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            d[u] = dy[(m + u * RL) * ldy];
            xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x + (m + u * RL) * ldx + chunk * 4));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += d[u] * xv[u];
            dsum += d[u];
        }
    }
    for (; m < r1; m += RL) {
        const float d = dy[m * ldy];
        acc += d * *reinterpret_cast<const f32x4 *>(x + m * ldx + chunk * 4);
        dsum += d;
    }
    *reinterpret_cast<f32x4 *>(&red[(rlane * cpr + chunk) * 4]) = acc;
    if (chunk != 0) dsum = 0.f;  // every row lane saw each dv once per chunk: count it once
    dsum = wave_sum64(dsum);
    if ((threadIdx.x & 63) == 0) dsum_s[threadIdx.x >> 6] = dsum;
    __syncthreads();
    // with a workspace the block's sums go to part[block][4 cpr + 4] and gemv_dw_reduce_kernel adds the blocks in order
    // (bit-reproducible); without one they are added atomically
    float *mine = part ? part + (size_t)blockIdx.x * (cpr * 4 + 4) : nullptr;
    for (int k = threadIdx.x; k < cpr * 4 && k < in; k += 256) {
        float s = 0.f;
        for (int r = 0; r < RL; ++r) s += red[r * cpr * 4 + k];
        if (mine) mine[k] = s;
        else atomicAdd(dw + k, s);
    }
    if (threadIdx.x == 0) {
        const float ds = dsum_s[0] + dsum_s[1] + dsum_s[2] + dsum_s[3];
        if (mine) mine[cpr * 4] = ds;
        else if (db) atomicAdd(db, ds);
    }
}

// block = 16 columns x 16 row lanes; a thread adds every 16th block partial with 16 loads in flight, the 16 lanes of a column
// meet in LDS and are added in lane order: a fixed summation order
__global__ __launch_bounds__(256) void gemv_dw_reduce_kernel(const float *__restrict__ part, int blocks, int kp,
                                                             float *__restrict__ dw, float *__restrict__ db, int in) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + cl;  // columns 0..in-1, and column kp = the bias gradient
    const bool live = k < in || k == kp;
    const int stride = kp + 4;
    float a16[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a16[u] = 0.f;
    if (live) {
        int b = rl;
        for (; b + 15 * 16 < blocks; b += 16 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(b + 16 * u) * stride + k];
#pragma unroll
            for (int u = 0; u < 16; ++u) a16[u] += v[u];
        }
        for (int u = 0; b < blocks; b += 16, ++u) a16[u] += part[(size_t)b * stride + k];
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += a16[u];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][cl];
        if (k == kp) {
            if (db) atomicAdd(db, t);
        } else
            atomicAdd(dw + k, t);
    }
}

// the shapes the three kernels take: padded input width a power of two in [32, 1024] (one 16-byte chunk per thread column)
bool gemv_head_ok(int out, int kp) { return out == 1 && kp >= 32 && kp <= 1024 && (kp & (kp - 1)) == 0; }

int launch_gemv_fwd(hipStream_t st, const float *x, int64_t ldx, const float *w, const float *b, float *y, int64_t ldy,
                    int64_t n, int kp, int pout) {
    if (n <= 0) return 0;
    const int64_t blocks = cdiv(n, 16);
    hipLaunchKernelGGL(gemv_fwd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, ldx, w, b, y, ldy, n, kp, pout);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_gemv_dx(hipStream_t st, const float *dy, int64_t ldy, const float *w, const float *mask, int64_t ldm, float *dx,
                   int64_t ldc, int kp, int64_t n) {
    if (n <= 0) return 0;
    const int cpr = kp / 4;
    hipLaunchKernelGGL(gemv_dx_kernel, dim3((unsigned)cdiv(n * cpr, 256)), dim3(256), 0, st, dy, ldy, w, mask, ldm, dx, ldc, cpr, n);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// returns -1 when the bitmask form does not apply (width not a multiple of 128)
int launch_gemv_dx_bits(hipStream_t st, const float *dy, int64_t ldy, const float *w, const unsigned long long *bits, float *dx,
                        int64_t ldc, int kp, int64_t n) {
    if (n <= 0) return 0;
    if (!bits || kp % 128 != 0) return -1;
    hipLaunchKernelGGL(gemv_dx_bits_kernel, dim3((unsigned)cdiv(n, 128), (unsigned)(kp / 128)), dim3(256), 0, st, dy, ldy, w, bits,
                       dx, ldc, n);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

int launch_gemv_dw(hipStream_t st, const float *dy, int64_t ldy, const float *x, int64_t ldx, float *dw, float *db, int in,
                   int kp, int64_t n, float *ws, size_t ws_floats) {
    if (n <= 0) return 0;
    // 1024 blocks at M = 65,536: enough loads in flight to stream h at HBM rate; larger M keeps about 2048 blocks
    const int rows_per_block = n > 131072 ? (int)round_up(cdiv(n, 2048), 64) : 64;
    const int blocks = (int)cdiv(n, rows_per_block);
    float *part = (ws && ws_floats >= (size_t)blocks * (kp + 4)) ? ws : nullptr;
    hipLaunchKernelGGL(gemv_dw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, ldy, x, ldx, dw, db, in, kp / 4, n,
                       rows_per_block, part);
    RLPPO_LAUNCH_CHECK();
    if (part) {
        hipLaunchKernelGGL(gemv_dw_reduce_kernel, dim3((unsigned)cdiv(kp + 1, 16)), dim3(256), 0, st, part, blocks, kp, dw, db, in);
        RLPPO_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace rlppo
