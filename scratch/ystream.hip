// scratch: how fast can a K1-shaped grid stream Y with K1's access pattern?  (no compute)
// mode 0: 4 waves/WG load 32x32 tiles as 16 dword loads (accumulator layout), depth D slots ahead
// mode 1: 8 waves/WG each load a 16x32 half tile (8 dword loads)
// mode 2: 4 waves, dwordx4 row-contiguous (8 rows x 128 B per instruction, 4 instr per 32x32 tile)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int DEPTH, int NMFMA = 0>
__global__ __launch_bounds__(512, 2) void k(const float* Y, int64_t ld, int M, int N, int RP, int gridX, float* out) {
    f32x16 c0 = {}, c1 = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)threadIdx.x; fb[i] = (__bf16)1.0f; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int lin = blockIdx.x, xcd = lin & 7, idx = lin >> 3, gy = N / 256;
    const int rowRegion = idx % gridX, colRegion = xcd * (gy >> 3) + idx / gridX;
    const int row0 = rowRegion * RP * 128, col0 = colRegion * 256;
    const int T = RP * 8;
    float acc = 0.f;
    float y[DEPTH][16] = {};
    unsigned long long t0 = __builtin_readcyclecounter();
    auto issue = [&](float (&d)[16], int t) {
        const int tc = t < T ? t : T - 1;
        const int rp = tc / 8, cb = tc % 8;
        if (MODE == 0) {
            if (w < 4) {
                const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + 4 * hi) * ld + col0 + cb * 32 + l31;
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = src[(int64_t)((i & 3) + 8 * (i >> 2)) * ld];
            }
        } else if (MODE == 1) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 16 + 2 * hi) * ld + col0 + cb * 32 + l31;
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = src[(int64_t)((i & 1) + 4 * (i >> 1)) * ld];
        } else if (MODE == 3) {
        } else {
            if (w < 4) {
                const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + (lane >> 3)) * ld + col0 + cb * 32 + (lane & 7) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)i * 8 * ld);
                    d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
                }
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(y[d], d);
    for (int t = 0; t < T; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += y[d][i];
            issue(y[d], t + d + DEPTH);
#pragma unroll
            for (int q = 0; q < NMFMA / 2; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, c1, 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    if (acc == 123.456f) out[0] = acc + c0[0] + c1[3];
    if (blockIdx.x == 0 && tid == 0) out[1] = (float)(__builtin_readcyclecounter() - t0);
}
int main() {
    const int M = 16384, N = 16384, RP = 16, gridX = 8;
    float* Y; float* out;
    CHECK(hipMalloc(&Y, (size_t)M * N * 4)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(Y, 0, (size_t)M * N * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(gridX * (N / 256)), dim3(512), 0, 0, Y, (int64_t)N, M, N, RP, gridX, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float hcyc[2]; hipMemcpy(hcyc, out, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-28s %.3f ms  %.0f GB/s   WG0 cycles %.0f\n", name, ms / 10, (double)M * N * 4 / (ms / 10) / 1e6, hcyc[1]);
        }
    };
    run("dword 4 waves depth1", k<0, 1>);
    run("dword 4 waves depth2", k<0, 2>);
    run("dword 4 waves depth4", k<0, 4>);
    run("no loads, 24 mfma/wave", k<3, 1, 24>);
    run("no loads, 48 mfma/wave", k<3, 1, 48>);
    run("dword d1 + 24 mfma/wave", k<0, 1, 24>);
    run("dword d2 + 24 mfma/wave", k<0, 2, 24>);
    run("dword d1 + 48 mfma/wave", k<0, 1, 48>);
    run("dword d2 + 48 mfma/wave", k<0, 2, 48>);
    run("dwordx4 4 waves depth1", k<2, 1>);
    run("dwordx4 4 waves depth2", k<2, 2>);
    run("dwordx4 4 waves depth4", k<2, 4>);
    return 0;
}
