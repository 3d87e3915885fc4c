// scratch (GPU): what does a grid barrier over 256 resident workgroups of 1024 threads cost on MI355X, by protocol?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gb scratch/r5_grid_barrier_bench.hip && /tmp/gb
// A: k_ada_tail's barrier (8 group counters -> top counter -> 8 published generation words, waiters poll their group's word)
// B: one counter, everybody polls the counter itself
// C: one counter, the last arrival publishes ONE generation word, everybody polls that word
// D: one counter, the last arrival publishes 8 generation words (one per blockIdx & 7 = per XCD), waiters poll theirs
// E: as D, the 8 words 64 bytes apart in ONE 512-byte block vs (D) 4 KB apart
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Bar { unsigned cnt[8][1024]; unsigned top[1024]; unsigned gen[8][1024]; unsigned one[1024]; unsigned genc[8][16]; };
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int V, int SLEEP>
__global__ __launch_bounds__(1024) void k(Bar* b, int n, float* sink) {
    float acc = 0.f;
    const unsigned g = blockIdx.x & 7, per = gridDim.x / 8;
    for (int i = 1; i <= n; ++i) {
        const unsigned e = (unsigned)i;
        acc += (float)threadIdx.x * 1e-9f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (V == 0) {
                if (add(&b->cnt[g][0]) + 1 == per * e) {
                    if (add(&b->top[0]) + 1 == 8 * e) for (unsigned q = 0; q < 8; ++q) st(&b->gen[q][0], e);
                }
                while (ld(&b->gen[g][0]) < e) __builtin_amdgcn_s_sleep(SLEEP);
            } else if (V == 1) {
                add(&b->one[0]);
                while (ld(&b->one[0]) < gridDim.x * e) __builtin_amdgcn_s_sleep(SLEEP);
            } else if (V == 2) {
                if (add(&b->one[0]) + 1 == gridDim.x * e) st(&b->gen[0][0], e);
                while (ld(&b->gen[0][0]) < e) __builtin_amdgcn_s_sleep(SLEEP);
            } else if (V == 3) {
                if (add(&b->one[0]) + 1 == gridDim.x * e) for (unsigned q = 0; q < 8; ++q) st(&b->gen[q][0], e);
                while (ld(&b->gen[g][0]) < e) __builtin_amdgcn_s_sleep(SLEEP);
            } else {
                if (add(&b->one[0]) + 1 == gridDim.x * e) for (unsigned q = 0; q < 8; ++q) st(&b->genc[q][0], e);
                while (ld(&b->genc[g][0]) < e) __builtin_amdgcn_s_sleep(SLEEP);
            }
        }
        __syncthreads();
    }
    if (acc == 123.f) sink[0] = acc;
}
template <int V, int SLEEP>
static void run(const char* name, Bar* b, float* sink) {
    const int n = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(b, 0, sizeof(Bar)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<V, SLEEP>), dim3(256), dim3(1024), 0, 0, b, n, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("%-60s sleep %d: %.2f us per barrier\n", name, SLEEP, 1e3 * ms / n);
    }
}
int main() {
    Bar* b; float* sink; CK(hipMalloc(&b, sizeof(Bar))); CK(hipMalloc(&sink, 4));
    run<0, 4>("A hierarchical (k_ada_tail)", b, sink);
    run<0, 1>("A hierarchical (k_ada_tail)", b, sink);
    run<1, 4>("B one counter, polled itself", b, sink);
    run<1, 1>("B one counter, polled itself", b, sink);
    run<2, 4>("C one counter + one generation word", b, sink);
    run<2, 1>("C one counter + one generation word", b, sink);
    run<3, 4>("D one counter + 8 generation words (4 KB apart)", b, sink);
    run<3, 1>("D one counter + 8 generation words (4 KB apart)", b, sink);
    run<4, 4>("E one counter + 8 generation words (64 B apart)", b, sink);
    run<4, 1>("E one counter + 8 generation words (64 B apart)", b, sink);
    return 0;
}
