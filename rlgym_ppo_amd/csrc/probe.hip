// probe.hip -- diagnostic micro-kernels (not on the product path): what does the GEMM inner loop sustain when its
// ingredients are added one at a time?  64 MFMAs (2 x 8 accumulator blocks, the wave tile of every GEMM kernel here)
// per 16-deep k chunk, plus optionally the LDS fragment reads and/or the global fragment loads of that chunk.
#include "common.hpp"

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
