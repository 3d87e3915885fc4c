// scratch: cost of a grid-wide barrier on MI355X (cooperative launch): cooperative_groups grid.sync() vs a hand-rolled
// sense-reversing counter barrier with agent-scope fences.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(1024) void k_cg(int n, float* out) {
    cg::grid_group g = cg::this_grid();
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) { v += 1.f; g.sync(); }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}

__device__ __forceinline__ void my_grid_sync(unsigned* counter, unsigned nblocks, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        ++epoch;
        const unsigned target = epoch * nblocks;
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}
__global__ __launch_bounds__(1024) void k_my(int n, unsigned* counter, float* out) {
    unsigned epoch = 0;
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) { v += 1.f; my_grid_sync(counter, gridDim.x * gridDim.y, epoch); }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}

int main() {
    float* out; unsigned* counter;
    hipMalloc(&out, 4); hipMalloc(&counter, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512}) {
        for (int n : {1, 101}) {
            for (int which = 0; which < 2; ++which) {
                hipMemset(counter, 0, 4);
                void* args_cg[] = {&n, &out};
                void* args_my[] = {&n, &counter, &out};
                hipEventRecord(e0);
                hipError_t e = which == 0 ? hipLaunchCooperativeKernel((void*)k_cg, dim3(blocks), dim3(1024), args_cg, 0, nullptr)
                                          : hipLaunchCooperativeKernel((void*)k_my, dim3(blocks), dim3(1024), args_my, 0, nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("blocks %d syncs %d %s: %s %.1f us\n", blocks, n, which == 0 ? "cg::grid.sync" : "counter barrier", hipGetErrorString(e), ms * 1e3);
            }
        }
    }
    return 0;
}
