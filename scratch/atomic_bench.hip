// micro-benchmark: throughput of fp32 global atomic adds (no return) when many workgroups fold
// partial tiles into one 4 MiB accumulator, vs plain stores of the same volume.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_atomic(float* acc, int n_acc, int reps, int mode) {
    // each WG walks the accumulator with a WG-dependent offset; lanes contiguous (coalesced)
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = gridDim.x * blockDim.x;
    for (int r = 0; r < reps; ++r) {
        int idx = (tid + r * total + blockIdx.x * 4099) % n_acc;
        if (mode == 0) atomicAdd(&acc[idx], 1.0f);
        else if (mode == 1) __hip_atomic_fetch_add(&acc[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else acc[idx] = 1.0f;
    }
}
int main() {
    const int n_acc = 1 << 20;  // 4 MiB
    float* d; hipMalloc(&d, n_acc * 4); hipMemset(d, 0, n_acc * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        for (int grid : {256, 1024, 4096}) {
            const int reps = 64;
            k_atomic<<<grid, 256>>>(d, n_acc, reps, mode);
            hipDeviceSynchronize();
            hipEventRecord(a);
            k_atomic<<<grid, 256>>>(d, n_acc, reps, mode);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double ops = (double)grid * 256 * reps;
            printf("mode %d (%s) grid %d: %.3f ms, %.1f Gop/s, %.2f TB/s-equivalent\n", mode,
                   mode == 0 ? "atomicAdd agent" : mode == 1 ? "atomic workgroup-scope" : "plain store", grid, ms, ops / ms / 1e6, ops * 4 / ms / 1e9);
        }
    }
    return 0;
}
