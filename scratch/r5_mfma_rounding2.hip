// scratch [r5], second question: are small products truncated when they are aligned to a large addend?  C = 2^24 (ulp 2) + 16 products of 2^-s each (exact sum 2^(4-s)):
// an adder that keeps every product to full width gives RNE(2^24 + 2^(4-s)); one that truncates each aligned product to g guard bits loses them from s > g on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    for (int s = 0; s <= 14; ++s)
        for (int sign = 0; sign < 2; ++sign) {
            f16x8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = (_Float16)1.0f; b[q] = (_Float16)ldexpf(sign ? -1.f : 1.f, -s); }
            f32x16 c;
            for (int i = 0; i < 16; ++i) c[i] = 16777216.0f;
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            if (lane == 0) out[2 * s + sign] = c[0] - 16777216.0f;
        }
    // C = 0: a large product (2^24) in slot 0 and 15 products of 2^-s
    for (int s = 0; s <= 14; ++s) {
        f16x8 a, b;
        for (int q = 0; q < 8; ++q) { const bool big = (lane >> 5) == 0 && q == 0; a[q] = (_Float16)(big ? 4096.0f : 1.0f); b[q] = (_Float16)(big ? 4096.0f : ldexpf(1.f, -s)); }
        f32x16 c;
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (lane == 0) out[64 + s] = c[0] - 16777216.0f;
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4); (void)hipMemset(d, 0, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s <= 14; ++s)
        printf("16 x 2^-%-2d (sum %8.4f): C=2^24 + -> %+.0f   C=2^24 - -> %+.0f   | 2^24 as a product + 15 x 2^-%-2d (sum %8.4f) -> %+.0f\n", s, ldexp(16.0, -s), h[2 * s], h[2 * s + 1], s, 15 * ldexp(1.0, -s), h[64 + s]);
    return 0;
}
