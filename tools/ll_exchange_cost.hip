// Cost of exchanging a 16 x 256 fp32 activation tile between G workgroups INSIDE one launch (round-5 question: can the rollout
// step at <= 16 rows be split by output columns over 16 CUs?  One CU is MFMA-bound at 3.4 us per 256 x 256 layer at 16 rows.)
//   form 0: flag-in-data ("LL") -- every value travels as one 8-byte relaxed agent-scope atomic {bits, flag}; a reader spins on the data.
//   form 1: plain stores + release fence + counter; readers spin on the counter, acquire, load.
// stride 1: consecutive workgroups (one per XCD, round robin); stride 8: only ids that are multiples of 8 work (same XCD if the
// dispatcher deals ids round robin).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ll tools/ll_exchange_cost.hip ; run: /tmp/ll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int SPIN_MAX = 1 << 22;

__global__ void __launch_bounds__(256) exch(uint64_t* buf, float* plain, unsigned* counters, unsigned* seqp, int G, int stride, int layers, int form,
                                            long long* t_out, int* err, float* sink) {
    if (blockIdx.x % stride) return;
    const int w = blockIdx.x / stride, tid = threadIdx.x;
    if (w >= G) return;
    const unsigned seq = *seqp;
    long long t0 = wall_clock64();
    float v = float(tid) * 1e-3f + float(w);
    for (int l = 0; l < layers; ++l) {
        const unsigned flag = seq * 8u + unsigned(l) + 1u;
        float sum = 0.f;
        if (form == 0) {
            uint64_t pack = (uint64_t(flag) << 32) | uint64_t(__float_as_uint(v));
            __hip_atomic_store(&buf[size_t(l) * 4096 + w * 256 + tid], pack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // all G loads of a thread in flight together; retry until every flag matches
            uint64_t got[16]; int spins = 0;
            while (true) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < G) got[j] = __hip_atomic_load(&buf[size_t(l) * 4096 + j * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < G) ok &= unsigned(got[j] >> 32) == flag;
                if (ok) break;
                if (++spins > SPIN_MAX) { *err = 1; break; }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < G) sum += __uint_as_float(unsigned(got[j]));
        } else {
            plain[size_t(l) * 4096 + w * 256 + tid] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&counters[l], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(&counters[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (seq + 1u) * unsigned(G)) {
                    if (++spins > SPIN_MAX) { *err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int j = 0; j < G; ++j) sum += plain[size_t(l) * 4096 + j * 256 + tid];
        }
        v = sum * 1e-3f;
    }
    long long t1 = wall_clock64();
    sink[w * 256 + tid] = v;
    if (w == 0 && tid == 0) { *t_out = t1 - t0; *seqp = seq + 1u; }
}

int main() {
    uint64_t* buf; float* plain; unsigned *counters, *seqp; long long* t_out; int* err; float* sink;
    CK(hipMalloc(&buf, 8 * 4096 * sizeof(uint64_t))); CK(hipMemset(buf, 0, 8 * 4096 * sizeof(uint64_t)));
    CK(hipMalloc(&plain, 8 * 4096 * sizeof(float)));
    CK(hipMalloc(&counters, 8 * sizeof(unsigned)));
    CK(hipMalloc(&seqp, sizeof(unsigned)));
    CK(hipMalloc(&err, sizeof(int))); CK(hipMemset(err, 0, sizeof(int)));
    CK(hipMalloc(&sink, 16 * 256 * sizeof(float)));
    CK(hipHostMalloc(&t_out, sizeof(long long)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
    printf("wall clock %d kHz\n", wc_khz);
    const int reps = 200;
    for (int form = 0; form < 2; ++form)
        for (int stride : {1, 8})
            for (int G : {4, 8, 16})
                for (int layers : {0, 1, 3, 7}) {
                    CK(hipMemset(counters, 0, 8 * sizeof(unsigned))); CK(hipMemset(seqp, 0, sizeof(unsigned)));
                    CK(hipMemset(buf, 0, 8 * 4096 * sizeof(uint64_t)));
                    std::vector<double> us;
                    for (int r = 0; r < reps + 20; ++r) {
                        hipLaunchKernelGGL(exch, dim3(G * stride), dim3(256), 0, 0, buf, plain, counters, seqp, G, stride, layers, form, t_out, err, sink);
                        CK(hipDeviceSynchronize());
                        if (r >= 20) us.push_back(double(*t_out) * 1e3 / wc_khz);
                    }
                    std::sort(us.begin(), us.end());
                    // back-to-back launches, event timed
                    CK(hipEventRecord(e0, 0));
                    for (int r = 0; r < reps; ++r)
                        hipLaunchKernelGGL(exch, dim3(G * stride), dim3(256), 0, 0, buf, plain, counters, seqp, G, stride, layers, form, t_out, err, sink);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    int herr = 0; CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
                    printf("form %d stride %d G %2d layers %d: in-kernel (workgroup 0) median %6.2f us  min %6.2f  p90 %6.2f ; back-to-back %6.2f us per launch%s\n",
                           form, stride, G, layers, us[us.size() / 2], us[0], us[us.size() * 9 / 10], ms * 1e3 / reps, herr ? "  SPIN TIMEOUT" : "");
                    if (herr) return 2;
                }
    return 0;
}
