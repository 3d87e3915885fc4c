// scratch [r5]: how does v_mfma_f32_32x32x16_f16 round?  C = +-2^24 (ulp 2) plus n unit products (A = 1, B = 1 in n of the 16 k slots), and a big product + n unit
// products inside ONE instruction with C = 0; the same for the exact-fp32 MFMA (32x32x2) and for a plain v_fma_f32 chain.  RNE gives 2^24 + {0,2,4,4,4,...} for n = 1,2,3,
// truncation (toward zero) 2^24 + {0,2,2,4,4}.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    int o = 0;
    for (int sign = 1; sign >= -1; sign -= 2)
        for (int n = 1; n <= 7; ++n) {
            f16x8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = (_Float16)1.0f; b[q] = (_Float16)((lane >> 5) == 0 && q < n ? (float)sign : 0.0f); }   // k = 8 (lane >> 5) + q: n unit products
            f32x16 c;
            for (int i = 0; i < 16; ++i) c[i] = sign * 16777216.0f;
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            if (lane == 0) out[o] = c[0];
            ++o;
            // inside one instruction: one product of 2^24 (2^12 x 2^12) + n unit products, C = 0
            for (int q = 0; q < 8; ++q) { a[q] = (_Float16)((lane >> 5) == 1 && q == 0 ? 4096.0f : 1.0f); b[q] = (_Float16)((lane >> 5) == 0 ? (q < n ? (float)sign : 0.0f) : (q == 0 ? sign * 4096.0f : 0.0f)); }
            for (int i = 0; i < 16; ++i) c[i] = 0.f;
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            if (lane == 0) out[o] = c[0];
            ++o;
            // exact-fp32 MFMA: C = +-2^24, one product of n
            f32x16 d;
            for (int i = 0; i < 16; ++i) d[i] = sign * 16777216.0f;
            d = __builtin_amdgcn_mfma_f32_32x32x2f32((lane >> 5) == 0 ? (float)n : 0.f, (float)sign, d, 0, 0, 0);
            if (lane == 0) out[o] = d[0];
            ++o;
        }
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); hipMemset(d, 0, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int o = 0;
    for (int sign = 1; sign >= -1; sign -= 2)
        for (int n = 1; n <= 7; ++n) {
            printf("sign %+d n %d: f16 MFMA C=+-2^24 + n -> %+.0f | f16 MFMA (2^24 + n in one instr) -> %+.0f | f32 MFMA -> %+.0f   (exact %+d; RNE %+.0f)\n", sign, n,
                   h[o] - sign * 16777216.0f, h[o + 1] - sign * 16777216.0f, h[o + 2] - sign * 16777216.0f, sign * n, (double)((float)(sign * (16777216.0 + n)) - sign * 16777216.0f));
            o += 3;
        }
    return 0;
}
