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
// XT = float, or unsigned short for an activation stored as bf16 (the bf16 update precision): 4 values per lane and load.
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load4(const float *p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)); }
__device__ __forceinline__ f32x4 load4(const unsigned short *p) {
    const u16x4 h = __builtin_nontemporal_load(reinterpret_cast<const u16x4 *>(p));
    return f32x4{__uint_as_float((unsigned)h[0] << 16), __uint_as_float((unsigned)h[1] << 16), __uint_as_float((unsigned)h[2] << 16),
                 __uint_as_float((unsigned)h[3] << 16)};
}
template <typename XT>
__global__ __launch_bounds__(256) void gemv_fwd_kernel(const XT *__restrict__ x, int64_t ldx, const float *__restrict__ w,
                                                       const float *__restrict__ b, float *__restrict__ y, int64_t ldy,
                                                       int64_t n, int kp, int pout) {
    const int lane = threadIdx.x & 63;
    const float bias = b[0];
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; row < n; row += (int64_t)gridDim.x * 16) {
        const XT *xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xr[u] = x + (row + u < n ? row + u : row) * ldx;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 4; c < kp; c += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c);
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load4(xr[u] + c);
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

// The same product with the ReLU mask taken from the bitmask the hidden layer's forward GEMM wrote (csrc/gemm_detail.hpp,
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

// ------------------------------------------------------------------------------------ narrow heads, bf16 update precision
// In the bf16 update precision the output layer keeps its fp32 gradient dL/dout (the loss kernel's), but its input is a bf16
// activation and the gradient it hands back is a bf16 tensor.  For heads of at most 32 outputs (the critic's 1, the Gaussian
// policy's 16, the multi-discrete 21) both backward products are streaming, HBM-bound work: n_out multiply-adds per element
// moved.  thin_dx_b16: dxb[m][k] = round_bf16(sum_n dY[m][n] W[n][k]) masked by the hidden layer's ReLU bitmask (the thread map
// of gemv_dx_bits_kernel); thin_dw_b16: dW[n][k] += sum_m dY[m][n] hb[m][k], db[n] += sum_m dY[m][n] through per-lane partial
// sums and a fixed-order reduction (no atomics in the sums: bit-reproducible).
__global__ __launch_bounds__(256) void thin_dx_b16_kernel(const float *__restrict__ dy, int64_t ldy, int n_out,
                                                          const float *__restrict__ w, int64_t ldw,
                                                          const unsigned long long *__restrict__ bits,
                                                          unsigned short *__restrict__ dxb, int64_t ldc, int64_t n) {
    // block = one 128 x 128 tile of dX.  The tile's dY rows (128 x 32 floats) and the 256 bitmask words of the forward's lanes go
    // through LDS, so that a thread can own 8 rows x 8 CONSECUTIVE columns (16-byte bf16 stores, 256 contiguous bytes per row and
    // 16 lanes) whatever lane of the forward owned those elements: bit (8 i + j) * 4 + e of the word of lane
    // (wave = r >> 5, q = (c & 15) >> 2, r16 = r & 15), i = (r >> 4) & 1, j = c >> 4, e = c & 3, for row r and column c of the tile.
    __shared__ __attribute__((aligned(16))) float dys[128][32];
    __shared__ unsigned long long words[256];
    const int tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    words[tid] = bits[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 256 + tid];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // 128 rows x 8 chunks of 16 bytes
        const int idx = tid + 256 * u, r = idx >> 3, ch = idx & 7;
        const int64_t row = m0 + r < n ? m0 + r : n - 1;
        *reinterpret_cast<f32x4 *>(&dys[r][ch * 4]) = ch * 4 < n_out ? *reinterpret_cast<const f32x4 *>(dy + row * ldy + ch * 4)
                                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int cg = tid & 15, rl = tid >> 4, c0 = cg * 8;
    f32x4 o[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) o[u][0] = o[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < n_out; ++c) {
        const float *wr = w + (int64_t)c * ldw + n0 + c0;
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wr), w1 = *reinterpret_cast<const f32x4 *>(wr + 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float d = dys[rl + 16 * u][c];
            o[u][0] += d * w0;
            o[u][1] += d * w1;
        }
    }
    const int j = c0 >> 4, qa = (c0 & 15) >> 2;  // the two forward lanes (q = qa, qa + 1) that owned these 8 columns
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = rl + 16 * u;
        if (m0 + r >= n) continue;
        const int base = (r >> 5) * 64 + (r & 15), sh = (((r >> 4) & 1) * 8 + j) * 4;
        const unsigned ma = (unsigned)(words[base + qa * 16] >> sh) & 15u, mb = (unsigned)(words[base + (qa + 1) * 16] >> sh) & 15u;
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = o[u][0][e], x1 = o[u][1][e];
            h[e] = ((ma >> e) & 1u) ? __builtin_bit_cast(unsigned short, (__bf16)x0) : (unsigned short)0;
            h[4 + e] = ((mb >> e) & 1u) ? __builtin_bit_cast(unsigned short, (__bf16)x1) : (unsigned short)0;
        }
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 pk = u32x4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                               (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
        *reinterpret_cast<u32x4 *>(dxb + (m0 + r) * ldc + n0 + c0) = pk;
    }
}

// thread = (row lane, 8-byte column chunk of hb); NOUT accumulator rows of 4 columns each; eight rows in flight
// UNI: a row spans whole waves (cpr >= 64), so the row a wave works on is wave-uniform and its dY values arrive by scalar loads
// instead of n_out / 4 more vector loads per row (a vector-memory instruction costs its 16 address cycles even when all 64
// lanes ask for the same 16 bytes: with them the kernel ran at 1.1 TB/s).
template <int NOUT, bool UNI>
__global__ __launch_bounds__(256) void thin_dw_b16_kernel(const float *__restrict__ dy, int64_t ldy, const unsigned short *__restrict__ xb,
                                                          int64_t ldx, int cpr, int64_t n, int rows_per_block,
                                                          float *__restrict__ part) {
    const int chunk = threadIdx.x % cpr, RL = 256 / cpr;
    const int rlane = UNI ? __builtin_amdgcn_readfirstlane(threadIdx.x / cpr) : threadIdx.x / cpr;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    f32x4 acc[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;  // thread `chunk` < NOUT also sums column `chunk` of dY: the bias gradient
    const int dcol = chunk < NOUT ? chunk : 0;
    auto one_row = [&](int64_t m, const f32x4 xv) {
        const float *dr = dy + m * ldy;
#pragma unroll
        for (int c = 0; c < NOUT; c += 4) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dr + (NOUT >= 4 ? c : 0));
#pragma unroll
            for (int e = 0; e < 4 && c + e < NOUT; ++e) acc[c + e] += d[e] * xv;
        }
        dsum += dr[dcol];
    };
    int64_t m = r0 + rlane;
    for (; m + 7 * RL < r1; m += 8 * RL) {  // eight activation rows in flight per thread
        f32x4 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = load4(xb + (m + u * RL) * ldx + chunk * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) one_row(m + u * RL, xv[u]);
    }
    for (; m < r1; m += RL) one_row(m, load4(xb + m * ldx + chunk * 4));
    // part[(block * RL + rlane)][NOUT][4 cpr] then [NOUT] column sums
    const int kp = cpr * 4;
    float *mine = part + ((size_t)blockIdx.x * RL + rlane) * ((size_t)NOUT * kp + NOUT);
#pragma unroll
    for (int c = 0; c < NOUT; ++c) *reinterpret_cast<f32x4 *>(mine + (size_t)c * kp + chunk * 4) = acc[c];
    if (chunk < NOUT) mine[(size_t)NOUT * kp + chunk] = dsum;
}

// element e of a partial: e < nout_p * kp -> dW[e / kp][e % kp], else db[e - nout_p * kp]; block = 16 elements x 16 partial lanes,
// a lane adds every 16th partial with 16 loads in flight, the 16 lanes meet in LDS in lane order: a fixed summation order
__global__ __launch_bounds__(256) void thin_dw_reduce_kernel(const float *__restrict__ part, int lanes, int nout_p, int kp,
                                                             float *__restrict__ dw, float *__restrict__ db, int out, int in) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int stride = nout_p * kp + nout_p;
    const int e = blockIdx.x * 16 + cl;
    int r = 0, k = 0;
    bool live = false, is_db = false;
    if (e < nout_p * kp) {
        r = e / kp;
        k = e - r * kp;
        live = r < out && k < in;
    } else if (e < stride) {
        r = e - nout_p * kp;
        is_db = true;
        live = r < out && db != nullptr;
    }
    float a16[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a16[u] = 0.f;
    if (live) {
        int b = rl;
        for (; b + 15 * 16 < lanes; b += 16 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(b + 16 * u) * stride + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) a16[u] += v[u];
        }
        for (int u = 0; b < lanes; b += 16, ++u) a16[u] += part[(size_t)b * stride + e];
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += a16[u];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][cl];
        if (is_db) atomicAdd(db + r, t);  // one add per element and launch (accumulation on top of the arena's value)
        else atomicAdd(dw + (size_t)r * in + k, t);
    }
}

// the shapes the three kernels take: padded input width a power of two in [32, 1024] (one 16-byte chunk per thread column)
bool gemv_head_ok(int out, int kp) { return out == 1 && kp >= 32 && kp <= 1024 && (kp & (kp - 1)) == 0; }

int launch_gemv_fwd(hipStream_t st, const float *x, int64_t ldx, const float *w, const float *b, float *y, int64_t ldy,
                    int64_t n, int kp, int pout) {
    if (n <= 0) return 0;
    const int64_t blocks = cdiv(n, 16);
    hipLaunchKernelGGL(gemv_fwd_kernel<float>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, ldx, w, b, y, ldy, n, kp, pout);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
int launch_gemv_fwd_b16(hipStream_t st, const unsigned short *x, int64_t ldx, const float *w, const float *b, float *y, int64_t ldy,
                        int64_t n, int kp, int pout) {
    if (n <= 0) return 0;
    const int64_t blocks = cdiv(n, 16);
    hipLaunchKernelGGL(gemv_fwd_kernel<unsigned short>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x, ldx, w, b,
                       y, ldy, n, kp, pout);
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

// ---- narrow heads of the bf16 update precision
bool thin_head_ok(int out, int kp) { return out >= 1 && out <= 32 && kp >= 32 && kp <= 1024 && (kp & (kp - 1)) == 0; }
static int thin_nout(int out) { return out <= 1 ? 1 : out <= 8 ? 8 : out <= 16 ? 16 : out <= 24 ? 24 : 32; }
static int thin_rows_per_block(int64_t n, int RL) {
    const int64_t per_lane = n > 131072 ? round_up(cdiv(n, 2048), 8) : 64;  // about 2048 partial lanes at large M
    return (int)(per_lane * RL);
}
size_t thin_dw_ws_floats(int out, int kp, int64_t n) {
    if (!thin_head_ok(out, kp) || n <= 0) return 0;
    const int RL = 256 / (kp / 4) > 0 ? 256 / (kp / 4) : 1;
    const int64_t blocks = cdiv(n, thin_rows_per_block(n, RL));
    return (size_t)blocks * RL * ((size_t)thin_nout(out) * kp + thin_nout(out));
}
// returns -1 when the form does not apply
int launch_thin_dx_b16(hipStream_t st, const float *dy, int64_t ldy, int out, const float *w, int64_t ldw,
                       const unsigned long long *bits, unsigned short *dxb, int64_t ldc, int kp, int64_t n) {
    if (n <= 0) return 0;
    if (!bits || kp % 128 != 0 || out < 1 || out > 32 || ldc % 8 != 0 || ldy % 4 != 0 || ldw % 4 != 0) return -1;
    hipLaunchKernelGGL(thin_dx_b16_kernel, dim3((unsigned)cdiv(n, 128), (unsigned)(kp / 128)), dim3(256), 0, st, dy, ldy, out, w, ldw,
                       bits, dxb, ldc, n);
    RLPPO_LAUNCH_CHECK();
    return 0;
}
int launch_thin_dw_b16(hipStream_t st, const float *dy, int64_t ldy, const unsigned short *xb, int64_t ldx, float *dw, float *db,
                       int out, int in, int kp, int64_t n, float *ws, size_t ws_floats) {
    if (n <= 0) return 0;
    if (!thin_head_ok(out, kp) || ldx % 4 != 0 || ldy % 4 != 0) return -1;
    const int cpr = kp / 4, RL = 256 / cpr > 0 ? 256 / cpr : 1, nout_p = thin_nout(out);
    RLPPO_CHECK_ARG(ldy >= nout_p || nout_p == 1, "thin_dw: dY rows of %ld floats are narrower than the %d padded outputs", (long)ldy, nout_p);
    if (!ws || ws_floats < thin_dw_ws_floats(out, kp, n)) {
        set_error("thin_dw: workspace %zu < %zu floats", ws ? ws_floats : (size_t)0, thin_dw_ws_floats(out, kp, n));
        return RLPPO_ERR_WORKSPACE;
    }
    const int rows_per_block = thin_rows_per_block(n, RL);
    const int blocks = (int)cdiv(n, rows_per_block);
#define THIN_DW(NO)                                                                                                               \
    do {                                                                                                                          \
        if (cpr >= 64)                                                                                                            \
            hipLaunchKernelGGL((thin_dw_b16_kernel<NO, true>), dim3((unsigned)blocks), dim3(256), 0, st, dy, ldy, xb, ldx, cpr, n,   \
                               rows_per_block, ws);                                                                               \
        else                                                                                                                      \
            hipLaunchKernelGGL((thin_dw_b16_kernel<NO, false>), dim3((unsigned)blocks), dim3(256), 0, st, dy, ldy, xb, ldx, cpr, n,  \
                               rows_per_block, ws);                                                                               \
    } while (0)
    switch (nout_p) {
        case 1: THIN_DW(1); break;
        case 8: THIN_DW(8); break;
        case 16: THIN_DW(16); break;
        case 24: THIN_DW(24); break;
        default: THIN_DW(32); break;
    }
#undef THIN_DW
    RLPPO_LAUNCH_CHECK();
    const int elems = nout_p * kp + nout_p;
    hipLaunchKernelGGL(thin_dw_reduce_kernel, dim3((unsigned)cdiv(elems, 16)), dim3(256), 0, st, ws, blocks * RL, nout_p, kp, dw, db,
                       out, in);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo
