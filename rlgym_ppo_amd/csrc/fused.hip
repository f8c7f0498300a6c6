// fused.hip -- layer-fused MLP chains for the 256-wide networks (BASELINE configs[1]: 256x3 policy and critic).
//
// Why: the layer-by-layer GEMMs of gemm.hip sit near the ridge of the roofline (hidden layer: 8.6 GFLOP against
// 200-270 MB of HBM traffic = 32-43 flop/B; the machine balance is 157 TFLOP/s / ~6.3 TB/s = 25 flop/B), so MFMA and
// HBM are both ~60 % busy and neither can be pushed while the other is (profiles/r01_pmc_*_v2.csv), and every launch
// pays ~10 us of ramp-up/tail at only 50-100 us of work.  Here one workgroup carries a tile of 128 rows through ALL
// layers: the activation tile lives in LDS (128 KB, XOR-swizzled), each layer's output is written to HBM once (the
// backward pass needs it) and never read back by the forward chain; weights (<= 256 KB per layer, L2-resident) go
// straight from global memory into MFMA fragments.  Forward chain traffic per net drops from ~800 MB to ~235 MB and
// 4 launches become 1.
//
//   mlp_fwd_fused_kernel : obs tile -> h1 -> ... -> hL -> head          (bias+ReLU epilogues, bias(+tanh) on the head)
//   mlp_bwd_fused_kernel : dOut tile -> dhL -> ... -> dh1               (ReLU-mask epilogues; W^T operands)
//
// Geometry: 512 threads = 8 waves as 4 row groups (32 rows) x 2 column halves (128 of 256 columns); wave tile 32 x 128 =
// 2 x 8 MFMA blocks of 16x16 (v_mfma_f32_16x16x4_f32, exact fp32).  No barrier inside a layer (the activation tile is
// read-only while a layer runs); two barriers between layers.  One workgroup per CU (LDS), persistent over row tiles.
#include "common.hpp"

namespace rlppo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int FR = 128;   // rows per tile
constexpr int FH = 256;   // hidden width handled by the fused kernels
constexpr int FCH = FH / 4;  // 16-byte chunks per activation row in LDS

struct FusedLayer {
    const float *w;   // fwd: packed W[pout][pin] ; bwd: packed W^T[pin][pout]  (contraction-contiguous rows)
    const float *b;   // fwd: packed bias[pout]   ; bwd: unused
    float *out;       // fwd: activation h_l [M][256] (or the head output) ; bwd: dX_l [M][256]
    const float *mask;  // bwd: saved activation whose ReLU mask applies to out ([M][256])
    int k;            // contraction length (multiple of 16, <= 256)
    int n;            // padded output width (256 for hidden layers; 32/64/96/128 for a head)
};

struct FusedArgs {
    const float *in;        // fwd: observation rows [*][ld_in] ; bwd: dOut rows [M][ld_in]
    int64_t ld_in;
    const int64_t *idx;     // fwd: minibatch gather (may be null)
    int64_t M;
    int k_in;               // valid (padded) width of `in` rows, multiple of 16, <= 256
    int n_layers;           // layers in `L`
    int out_tanh;           // fwd: tanh on the last layer
    int n_tiles;
    FusedLayer L[RLPPO_MAX_LAYERS];
};

__device__ __forceinline__ int act_off(int row, int chunk) { return row * FCH + (chunk ^ (row & 15)); }

// One layer on the resident tile: acc[i][j] += Act[rows of this wave][0:K] . W[n_base + 16 j + r16][0:K]^T
template <int NBW>
__device__ __forceinline__ void layer_mma(const f32x4 *__restrict__ Act4, const float *__restrict__ W, int ldw, int K,
                                          int n_base, int row0, int r16, int q, f32x4 (&acc)[2][NBW]) {
    const int nkc = K >> 4;
    const float *wp[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) wp[j] = W + (int64_t)(n_base + j * 16 + r16) * ldw + q * 4;
    f32x4 fb[NBW], fbn[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(wp[j]);
    for (int kc = 0; kc < nkc; ++kc) {
        const int kn = kc + 1 < nkc ? kc + 1 : kc;  // last iteration re-reads its own chunk (harmless, L1 hit)
#pragma unroll
        for (int j = 0; j < NBW; ++j) fbn[j] = *reinterpret_cast<const f32x4 *>(wp[j] + kn * 16);
        f32x4 fa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = Act4[act_off(row0 + i * 16 + r16, kc * 4 + q)];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[i][j] = MFMA16(fb[j][s], fa[i][s], acc[i][j]);
#pragma unroll
        for (int j = 0; j < NBW; ++j) fb[j] = fbn[j];
    }
}

// Loads a [128][k_in] tile of `in` (optionally gathered) into the swizzled LDS activation tile.
__device__ __forceinline__ void load_input_tile(f32x4 *Act4, const FusedArgs &a, int64_t m0, int tid) {
    const int cpr = a.k_in >> 2;              // chunks per row
    const int total = FR * cpr;               // <= 128 * 64 = 8192 chunks, 16 per thread at most
    for (int base = tid; base < total; base += 512 * 4) {
        f32x4 v[4];
        int rr[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = base + u * 512;
            rr[u] = id / cpr;
            cc[u] = id - rr[u] * cpr;
            if (id < total) {
                int64_t m = m0 + rr[u];
                if (m >= a.M) m = a.M - 1;
                const int64_t src = a.idx ? a.idx[m] : m;
                v[u] = *reinterpret_cast<const f32x4 *>(a.in + src * a.ld_in + cc[u] * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * 512 < total) Act4[act_off(rr[u], cc[u])] = v[u];
    }
}

template <int HNB, bool BWD>
__global__ __launch_bounds__(512) void mlp_fused_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) float ActF[FR * FH];  // 128 KB
    f32x4 *Act4 = reinterpret_cast<f32x4 *>(ActF);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int rg = wave >> 1, ch = wave & 1;
    const int row0 = rg * 32;

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t m0 = (int64_t)tile * FR;
        load_input_tile(Act4, a, m0, tid);
        __syncthreads();

        const int n_hidden = BWD ? a.n_layers : a.n_layers - 1;  // layers with 256 outputs handled by the wide path
        int K = a.k_in;
        for (int l = 0; l < n_hidden; ++l) {
            const FusedLayer &Ly = a.L[l];
            f32x4 acc[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            layer_mma<8>(Act4, Ly.w, K, K, ch * 128, row0, r16, q, acc);
            // epilogue: out to HBM; keep the result in registers until every wave has finished reading the tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t m = m0 + row0 + i * 16 + r16;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = ch * 128 + j * 16 + q * 4;
                    f32x4 v = acc[i][j];
                    if (BWD) {
                        f32x4 h = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (m < a.M) h = *reinterpret_cast<const f32x4 *>(Ly.mask + m * FH + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = h[e] > 0.f ? v[e] : 0.f;
                    } else {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(Ly.b + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = v[e] + bv[e];
                            v[e] = x > 0.f ? x : 0.f;
                        }
                    }
                    acc[i][j] = v;
                    if (m < a.M && Ly.out) *reinterpret_cast<f32x4 *>(Ly.out + m * FH + n) = v;
                }
            }
            const bool feeds_next = BWD ? (l + 1 < n_hidden) : true;
            if (feeds_next) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        Act4[act_off(row0 + i * 16 + r16, ch * 32 + j * 4 + q)] = acc[i][j];
                __syncthreads();
            }
            K = FH;
        }

        if (!BWD) {  // narrow head: 2 * HNB blocks of 16 columns, split over the two column-half waves
            const FusedLayer &Ly = a.L[a.n_layers - 1];
            f32x4 acc[2][HNB];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < HNB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            layer_mma<HNB>(Act4, Ly.w, K, K, ch * HNB * 16, row0, r16, q, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t m = m0 + row0 + i * 16 + r16;
#pragma unroll
                for (int j = 0; j < HNB; ++j) {
                    const int n = ch * HNB * 16 + j * 16 + q * 4;
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(Ly.b + n);
                    f32x4 v = acc[i][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e] + bv[e];
                        if (a.out_tanh) x = tanhf(x);
                        v[e] = x;
                    }
                    if (m < a.M) *reinterpret_cast<f32x4 *>(Ly.out + m * Ly.n + n) = v;
                }
            }
        }
        __syncthreads();  // the tile is overwritten by the next row tile's input
    }
}

// Off by default.  Measured (tools/time_fwd.py, tools/ab_update.py, 1x MI355X): the fused forward chain runs at 89 TFLOP/s
// (278-289 us per 65,536-row pass) against 91-100 TFLOP/s (247-271 us) for the four layer-wise launches, and the whole
// update is 15.1 ms/epoch fused vs 12.7 ms layer-wise: with weights streamed as per-wave fragment loads from L2 the
// inner loop sustains only ~117-133 TFLOP/s (tools/probe2.py: '+B frags global'), and one 128 KB-LDS workgroup per CU
// leaves nothing to hide the inter-layer barriers.  Kept (parity-tested) as the starting point for a version that
// stages weights through LDS; enable with rlppo_dbg_set(6, 1) / RLPPO_TUNE=6=1.
static int g_fused = 0;
void set_fused(int v) { g_fused = v; }
bool fused_enabled() { return g_fused != 0; }

// eligibility: every hidden layer 256 wide (padded), padded input <= 256, head padded width in {32, 64, 96, 128}
bool fused_eligible(const NetLayout &net, int64_t mb) {
    if (!g_fused || mb < 4 * FR || net.n_layers < 2) return false;
    if (net.L[0].pin > FH || net.L[0].pin % 16) return false;
    for (int l = 0; l + 1 < net.n_layers; ++l)
        if (net.L[l].pout != FH) return false;
    const int ph = net.L[net.n_layers - 1].pout;
    return ph == 32 || ph == 64 || ph == 96 || ph == 128;
}

static int fused_grid(int n_tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return n_tiles < 256 ? n_tiles : 256;
        cus = prop.multiProcessorCount;
    }
    return n_tiles < cus ? n_tiles : cus;
}

// acts[l] = output buffer of layer l (hidden layers: [mb][256]; last: [mb][pout_last])
int launch_fused_forward(hipStream_t st, const NetLayout &net, const float *packed, const float *obs, int64_t ld_obs,
                         const int64_t *idx, int64_t mb, int out_tanh, float *const *acts) {
    FusedArgs a;
    a.in = obs;
    a.ld_in = ld_obs;
    a.idx = idx;
    a.M = mb;
    a.k_in = net.L[0].pin;
    a.n_layers = net.n_layers;
    a.out_tanh = out_tanh;
    a.n_tiles = (int)cdiv(mb, FR);
    for (int l = 0; l < net.n_layers; ++l) {
        a.L[l].w = packed + net.L[l].off_w;
        a.L[l].b = packed + net.L[l].off_b;
        a.L[l].out = acts[l];
        a.L[l].mask = nullptr;
        a.L[l].k = net.L[l].pin;
        a.L[l].n = net.L[l].pout;
    }
    const dim3 grid(fused_grid(a.n_tiles)), block(512);
    switch (net.L[net.n_layers - 1].pout) {
        case 32: hipLaunchKernelGGL((mlp_fused_kernel<1, false>), grid, block, 0, st, a); break;
        case 64: hipLaunchKernelGGL((mlp_fused_kernel<2, false>), grid, block, 0, st, a); break;
        case 96: hipLaunchKernelGGL((mlp_fused_kernel<3, false>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((mlp_fused_kernel<4, false>), grid, block, 0, st, a); break;
    }
    RLPPO_LAUNCH_CHECK();
    return 0;
}

// Backward dX chain: dOut = acts[last] ([mb][pout_last], holds dL/d(out)); produces dX_l for l = last..1 into dx[l-1]
// ([mb][256]) with the ReLU mask of acts[l-1].
int launch_fused_backward(hipStream_t st, const NetLayout &net, const float *packed, int64_t mb, float *const *acts,
                          float *const *dx) {
    const int last = net.n_layers - 1;
    FusedArgs a;
    a.in = acts[last];
    a.ld_in = net.L[last].pout;
    a.idx = nullptr;
    a.M = mb;
    a.k_in = net.L[last].pout;
    a.n_layers = last;  // number of dX products: layers last..1
    a.out_tanh = 0;
    a.n_tiles = (int)cdiv(mb, FR);
    for (int s = 0; s < last; ++s) {
        const int l = last - s;  // dX_l = (dY_l . W_l) masked by relu'(acts[l-1])
        a.L[s].w = packed + net.L[l].off_wt;  // W^T [pin_l][pout_l]: row = input feature, contraction over pout_l
        a.L[s].b = nullptr;
        a.L[s].out = dx[l - 1];
        a.L[s].mask = acts[l - 1];
        a.L[s].k = net.L[l].pout;
        a.L[s].n = net.L[l].pin;
    }
    if (last == 0) return 0;
    const dim3 grid(fused_grid(a.n_tiles)), block(512);
    hipLaunchKernelGGL((mlp_fused_kernel<1, true>), grid, block, 0, st, a);
    RLPPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace rlppo
