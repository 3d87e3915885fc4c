// scratch (round 3): what does one s_barrier cost in a 512-thread workgroup, one workgroup per CU?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE 0: barrier only; 1: + 16 dependent VALU per wave; 2: + 8 ds_write_b64 + 8 ds_read_b128 per wave; 3: 12 MFMA chain in waves 0-3 only;
// 4: 16 v_pk_fma_f32; 5: 16 v_fma_f32 x2 (same flops unpacked); 6: 16 global_load_dword (L2-resident) issue only
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void kb(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float v = out[8 + (tid & 7)];
    float2 pv = {v, v + 1.f};
    f32x16 c0 = {};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (float)((tid * 7 + i * 13) & 255) - 0.1f); fb[i] = (_Float16)(0.002f * (float)((tid * 3 + i * 5) & 127) - 0.11f); }
    float ld[16] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v = v * 1.0001f + 0.5f;
        } else if constexpr (MODE == 2) {
            float4* p = reinterpret_cast<float4*>(smem) + tid;
#pragma unroll
            for (int q = 0; q < 8; ++q) { float2 x = {v, v}; reinterpret_cast<float2*>(smem + 65536)[tid + 512 * q] = x; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float4 x = p[512 * q]; v += x.x + x.w; }
        } else if constexpr (MODE == 3) {
            if (w < 4) {
#pragma unroll
                for (int q = 0; q < 12; ++q) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c0, 0, 0, 0);
            }
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv) : "v"(pv));
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(pv.x) : "v"(pv.y)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(pv.y) : "v"(pv.x)); }
        } else if constexpr (MODE == 6) {
            if (w < 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) ld[q] += src[(size_t)(blockIdx.x * 4096 + (it & 7) * 32 + q * 16384 + lane)];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v + pv.x + pv.y + c0[0];
    for (int q = 0; q < 16; ++q) s += ld[q];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && tid == 0) out[1] = (float)(t1 - t0) / (float)iters;
}
int main() {
    float* out; float* src;
    CHECK(hipMalloc(&out, 256)); CHECK(hipMemset(out, 0, 256));
    CHECK(hipMalloc(&src, 64u << 20)); CHECK(hipMemset(src, 0, 64u << 20));
    const int iters = 2000;
    auto run = [&](const char* name, auto kern, int threads, int lds) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, out, src, iters); CHECK(hipDeviceSynchronize()); }
        float h[2]; CHECK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
        printf("%-58s %7.1f cycles per iteration\n", name, h[1]);
    };
    run("barrier only, 8 waves", kb<0, 512>, 512, 128 * 1024);
    run("barrier only, 4 waves", kb<0, 256>, 256, 128 * 1024);
    run("barrier only, 16 waves", kb<0, 1024>, 1024, 128 * 1024);
    run("16 dependent v_fma + barrier, 8 waves", kb<1, 512>, 512, 128 * 1024);
    run("8 ds_write_b64 + 8 ds_read_b128 per wave + barrier", kb<2, 512>, 512, 128 * 1024);
    run("12-MFMA dependent chain (waves 0-3) + barrier", kb<3, 512>, 512, 128 * 1024);
    run("16 dependent v_pk_fma_f32 + barrier", kb<4, 512>, 512, 128 * 1024);
    run("32 dependent v_fma_f32 + barrier", kb<5, 512>, 512, 128 * 1024);
    run("16 global_load_dword (waves 0-3, L2) + barrier", kb<6, 512>, 512, 128 * 1024);
    return 0;
}
