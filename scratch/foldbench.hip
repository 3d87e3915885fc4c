// scratch: slab fold read pattern -- [slab][row][K] (stride 4 MB between the 64 terms of a row) vs [row][slab][K] (16 KB contiguous)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int K = 64, NS = 64, ROWS = 16384, BLOCKS = 256, THREADS = 1024;
template <bool INTERLEAVED>
__global__ __launch_bounds__(THREADS) void k_fold(const float* __restrict__ slab, float* __restrict__ out) {
    const int l32 = threadIdx.x & 31;
    const long hw = ((long)blockIdx.x * THREADS + threadIdx.x) >> 5, nhw = ((long)BLOCKS * THREADS) >> 5;
    for (long r = hw; r < ROWS; r += nhw) {
        float g0 = 0.f, g1 = 0.f;
        const float* p = INTERLEAVED ? slab + r * NS * K + l32 : slab + r * K + l32;
        const long stride = INTERLEAVED ? K : (long)ROWS * K;
        for (int i = 0; i < NS; i += 4) {
            float v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[u][0] = __builtin_nontemporal_load(p + u * stride); v[u][1] = __builtin_nontemporal_load(p + u * stride + 32); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { g0 += v[u][0]; g1 += v[u][1]; }
            p += 4 * stride;
        }
        out[r * K + l32] = g0; out[r * K + l32 + 32] = g1;
    }
}
int main() {
    float *slab, *out;
    hipMalloc(&slab, (size_t)NS * ROWS * K * 4); hipMalloc(&out, (size_t)ROWS * K * 4);
    hipMemset(slab, 0, (size_t)NS * ROWS * K * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            hipMemset(slab, 0, (size_t)NS * ROWS * K * 4);      // (re)writes the slabs: similar cache state for both
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (which) hipLaunchKernelGGL(k_fold<true>, dim3(BLOCKS), dim3(THREADS), 0, 0, slab, out);
                else hipLaunchKernelGGL(k_fold<false>, dim3(BLOCKS), dim3(THREADS), 0, 0, slab, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us per fold of 268 MB (%.2f TB/s)\n", which ? "[row][slab][K]" : "[slab][row][K]", ms * 100, 268.4e6 / (ms * 1e-4) / 1e12);
        }
    return 0;
}
