// scratch (round 3): can the K = 128 gA tiles be folded INSIDE K1, a few panels behind their writers, out of the
// Infinity Cache?
//   part 1: kernel W writes X bytes, kernel R reads them back: read bandwidth against X (is the memory-side cache
//           write-allocating, and up to what footprint?)
//   part 2: the protocol itself on K1's grid: 256 persistent workgroups = 2 row regions x 128 column regions; per "panel"
//           step every workgroup writes a 128 x 128 fp32 tile (64 KB) into its column region's slab, bumps the panel's
//           arrival counter, and LAG panels later reads row c of that panel from all 128 slabs (128 x 512 B) and writes
//           the sum.  A step is padded with a fixed amount of dependent VALU work so that it lasts about as long as a panel
//           of k_grad_f16_k128 (4 slots = ~5.6 us).  Time per step with / without the fold, with / without the writes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k_write(f32x4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) p[i] = f32x4{v, v, v, v};
}
__global__ __launch_bounds__(1024) void k_read(const f32x4* p, size_t n4, float* out) {
    f32x4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) s += __builtin_nontemporal_load(p + i);
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;
}

constexpr int K = 128, PR = 128, NREG = 128, NROWREG = 2, THREADS = 512;
struct PArgs {
    float* slab;          // [NREG][M][K]
    float* out;           // [M][K]
    unsigned* arrive;     // [NROWREG][panels]
    int M, panels;        // panels per row region
    int lag, pad;
    int doWrite, doFold, wt;
    unsigned base;        // arrival words count from base (monotonic over launches)
    unsigned* fault;
};
__device__ __forceinline__ void store_wt(float* p, f32x4 v) {      // write-through to the device coherence point
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_dev(const float* p) {        // bypass this XCD's L2 copy
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(THREADS) void k_proto(PArgs a) {
    __shared__ float red[THREADS / 64][K];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int rowRegion = idx % NROWREG, colRegion = xcd * (NREG / 8) + idx / NROWREG;
    const int row0 = rowRegion * a.panels * PR;
    float v = (float)tid * 1e-3f;
    for (int p = 0; p < a.panels + a.lag; ++p) {
        // "compute": pad dependent VALU ops per thread
        for (int q = 0; q < a.pad; ++q) v = v * 1.0001f + 0.5f;
        if (p < a.panels && a.doWrite) {
            // tile: 128 rows x 128 k = 4096 float4; 512 threads x 8
            float* dst = a.slab + ((size_t)colRegion * a.M + row0 + (size_t)p * PR) * K;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = (q * THREADS + tid) * 4;
                const f32x4 x = {v, v + 1.f, v + 2.f, (float)(colRegion + 1)};
                if (a.wt) store_wt(dst + e, x);
                else *reinterpret_cast<f32x4*>(dst + e) = x;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!a.wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.arrive + rowRegion * a.panels + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int pf = p - a.lag;
        if (pf >= 0 && a.doFold) {
            if (a.doWrite) {
                if (tid == 0) {
                    const unsigned want = a.base + NREG;
                    long spins = 0;
                    while (__hip_atomic_load(a.arrive + rowRegion * a.panels + pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > 2000000) { *a.fault = 1; break; }
                    }
                    if (spins > 0) atomicAdd(a.fault + 1, 1u);
                }
                __syncthreads();
                if (!a.wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            // row (prow + colRegion) of the panel from all 128 slabs: 128 x 512 B; wave w takes slabs w, w + 8, ...; lane: 32 lanes x 16 B = one slab's row, two slabs per instruction
            const size_t rowoff = ((size_t)row0 + (size_t)pf * PR + colRegion) * K;
            f32x4 s = {0, 0, 0, 0};
            f32x4 t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int sl = (q * 8 + w) * 2 + (lane >> 5);
                const float* src = a.slab + (size_t)sl * a.M * K + rowoff + (lane & 31) * 4;
                t[q] = a.wt ? load_dev(src) : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) s += t[q];
            // fold the two half-waves and the 8 waves through LDS
            s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32);
            if (lane < 32) *reinterpret_cast<f32x4*>(&red[w][lane * 4]) = s;
            __syncthreads();
            if (tid < K) {
                float r = 0.f;
#pragma unroll
                for (int i = 0; i < THREADS / 64; ++i) r += red[i][tid];
                a.out[rowoff + tid] = r;
            }
            __syncthreads();
        }
    }
    if (v == 123.456f) a.out[0] = v;
}

int main(int argc, char** argv) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float* out; CHECK(hipMalloc(&out, 1 << 20));
    // ---- part 1
    {
        const size_t maxB = (size_t)2048 << 20;
        float* buf; CHECK(hipMalloc(&buf, maxB));
        printf("part 1: write X then read X (256 workgroups x 1024)\n");
        for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
            const size_t n4 = (mb << 20) / 16;
            float tw = 0, tr = 0;
            for (int it = 0; it < 12; ++it) {
                float ms;
                CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_write, dim3(256), dim3(1024), 0, 0, (f32x4*)buf, n4, (float)it); CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 2) tw += ms;
                CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, 0, (const f32x4*)buf, n4, out); CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 2) tr += ms;
            }
            // read twice in a row (second read: whatever the first left in the caches)
            float tr2 = 0;
            for (int it = 0; it < 10; ++it) {
                float ms;
                hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, 0, (const f32x4*)buf, n4, out);
                CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, 0, (const f32x4*)buf, n4, out); CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); tr2 += ms;
            }
            printf("  X = %5zu MB: write %7.1f us (%6.2f TB/s)  read-after-write %7.1f us (%6.2f TB/s)  read-after-read %7.1f us (%6.2f TB/s)\n", mb,
                   tw * 100.f, (double)(mb << 20) / (tw / 10 * 1e-3) / 1e12, tr * 100.f, (double)(mb << 20) / (tr / 10 * 1e-3) / 1e12,
                   tr2 * 100.f, (double)(mb << 20) / (tr2 / 10 * 1e-3) / 1e12);
        }
        CHECK(hipFree(buf));
    }
    // ---- part 2
    {
        PArgs a{};
        a.panels = 32; a.M = NROWREG * a.panels * PR;   // 8192 rows
        CHECK(hipMalloc(&a.slab, (size_t)NREG * a.M * K * 4));
        CHECK(hipMemset(a.slab, 0, (size_t)NREG * a.M * K * 4));
        CHECK(hipMalloc(&a.out, (size_t)a.M * K * 4));
        CHECK(hipMalloc(&a.arrive, NROWREG * a.panels * 4)); CHECK(hipMemset(a.arrive, 0, NROWREG * a.panels * 4));
        CHECK(hipMalloc(&a.fault, 8)); CHECK(hipMemset(a.fault, 0, 8));
        unsigned base = 0;
        printf("part 2: 256 workgroups, %d panels each, tile 64 KB per panel and workgroup\n", a.panels);
        const int pads[] = {0, 400, 800};
        for (int pad : pads)
            for (int wt = 0; wt < 2; ++wt)
                for (int lag = 1; lag <= 4; ++lag) {
                    if (lag == 3) continue;
                    for (int cfg = 0; cfg < 4; ++cfg) {     // 0: nothing, 1: writes, 2: writes + fold, 3: fold only (no sync, reads whatever is there)
                        if (cfg != 2 && lag != 2) continue;
                        a.pad = pad; a.wt = wt; a.lag = lag;
                        a.doWrite = cfg == 1 || cfg == 2; a.doFold = cfg >= 2;
                        float tot = 0;
                        for (int it = 0; it < 8; ++it) {
                            a.base = base;
                            float ms;
                            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_proto, dim3(256), dim3(THREADS), 0, 0, a); CHECK(hipEventRecord(e1));
                            CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
                            if (a.doWrite) base += NREG;
                            if (it >= 2) tot += ms;
                        }
                        unsigned f[2]; CHECK(hipMemcpy(f, a.fault, 8, hipMemcpyDeviceToHost)); CHECK(hipMemset(a.fault, 0, 8));
                        std::vector<float> h((size_t)a.M * K);
                        CHECK(hipMemcpy(h.data(), a.out, h.size() * 4, hipMemcpyDeviceToHost));
                        // check: column 3 of every folded row = sum over regions of (region + 1) = 128 * 129 / 2 = 8256
                        long bad = 0;
                        if (cfg == 2) for (size_t r = 0; r < (size_t)a.M; ++r) if (h[r * K + 3] != 8256.f) ++bad;
                        printf("  pad %4d  %s  lag %d  %-12s: %7.1f us per launch = %6.2f us per panel   waits %u fault %u bad rows %ld\n", pad, wt ? "sc0sc1" : "plain ",
                               lag, cfg == 0 ? "nothing" : cfg == 1 ? "writes" : cfg == 2 ? "writes+fold" : "fold only", tot / 6 * 1e3f, tot / 6 * 1e3f / a.panels, f[1], f[0], bad);
                    }
                }
    }
    return 0;
}
