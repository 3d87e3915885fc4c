// scratch [r5]: does v8_split_pair (v_fma_mix*_f16) round like fp16(x) / fp16(x - h) to nearest-even?  And the mean error of an fp16 MFMA that adds 16 random-sign
// products far below a large accumulator (the consumers' situation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void v8_split_pair(float r0, float r1, float sc, unsigned& h, unsigned& l) {
    unsigned hh;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(r0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(r1), "v"(sc));
    h = hh;
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(r0), "v"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(r1), "v"(sc), "v"(h));
    l = lo;
}
__global__ void ksplit(const float* x, int n, float sc, unsigned* H, unsigned* L, unsigned* Href, unsigned* Lref) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h, l;
    v8_split_pair(x[2 * i], x[2 * i + 1], sc, h, l);
    H[i] = h; L[i] = l;
    _Float16 a0 = (_Float16)(x[2 * i] * sc), a1 = (_Float16)(x[2 * i + 1] * sc);
    _Float16 b0 = (_Float16)(x[2 * i] * sc - (float)a0), b1 = (_Float16)(x[2 * i + 1] * sc - (float)a1);
    Href[i] = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
    Lref[i] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
}
// one wave: acc = C (large) then `steps` MFMAs of random-sign products; out = acc[0..15] per lane; the host compares with the exact fp64 sums
__global__ void kmfma(const _Float16* A, const _Float16* B, float C, int steps, float* out) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = C;
    for (int s = 0; s < steps; ++s) {
        f16x8 a = *reinterpret_cast<const f16x8*>(A + ((size_t)s * 64 + lane) * 8), b = *reinterpret_cast<const f16x8*>(B + ((size_t)s * 64 + lane) * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = acc[i];
}
int main() {
    const int n = 1 << 20;
    float* hx = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = ((rand() / (float)RAND_MAX) - 0.5f) * ldexpf(1.f, (rand() % 12) - 6);
    float* dx; unsigned *dH, *dL, *dHr, *dLr;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dH, n * 2); (void)hipMalloc(&dL, n * 2); (void)hipMalloc(&dHr, n * 2); (void)hipMalloc(&dLr, n * 2);
    (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ksplit, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, 64.f, dH, dL, dHr, dLr);
    unsigned* H = (unsigned*)malloc(n * 2); unsigned* L = (unsigned*)malloc(n * 2); unsigned* Hr = (unsigned*)malloc(n * 2); unsigned* Lr = (unsigned*)malloc(n * 2);
    (void)hipMemcpy(H, dH, n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(L, dL, n * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(Hr, dHr, n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(Lr, dLr, n * 2, hipMemcpyDeviceToHost);
    int dh = 0, dl = 0;
    for (int i = 0; i < n / 2; ++i) { dh += H[i] != Hr[i]; dl += L[i] != Lr[i]; }
    printf("v8_split_pair against (_Float16) conversions: %d of %d high words differ, %d low words differ\n", dh, n / 2, dl);
    {   // ... and against the HOST's round-to-nearest-even conversions (the device pair could share a rounding mode that is not RNE)
        int bh = 0, bl = 0; double mean = 0, meanabs = 0;
        for (int i = 0; i < n; ++i) {
            const float v = hx[i] * 64.f;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const unsigned short dhw = (unsigned short)(H[i / 2] >> (16 * (i & 1))), dlw = (unsigned short)(L[i / 2] >> (16 * (i & 1)));
            bh += dhw != __builtin_bit_cast(unsigned short, h);
            bl += dlw != __builtin_bit_cast(unsigned short, l);
            const double rep = (double)(float)__builtin_bit_cast(_Float16, dhw) + (double)(float)__builtin_bit_cast(_Float16, dlw);
            mean += (rep - (double)v) / fabs((double)v + 1e-300); meanabs += fabs(rep - (double)v) / fabs((double)v + 1e-300);
        }
        printf("... against the host's RNE: %d of %d high halves differ, %d low halves differ; (h + l - x) / |x|: mean %+.3e, mean abs %.3e (2^-24 = 5.96e-08)\n", bh, n, bl, mean / n, meanabs / n);
    }
    // MFMA mean error
    for (int variant = 0; variant < 6; ++variant) {
        const int steps = variant == 5 ? 1 : 64;
        _Float16* hA = (_Float16*)malloc((size_t)steps * 64 * 8 * 2); _Float16* hB = (_Float16*)malloc((size_t)steps * 64 * 8 * 2);
        for (size_t i = 0; i < (size_t)steps * 64 * 8; ++i) {
            float a = (rand() / (float)RAND_MAX) * 2.f, b = (rand() / (float)RAND_MAX) * 2.f;
            if (variant == 1) b = b - 1.f;                    // random-sign products
            if (variant == 2) { a *= 64.f; b *= 64.f; }       // positive, products comparable with the accumulator's growth
            if (variant == 3) b = b - 1.f;                    // random-sign products on a NEGATIVE accumulator
            if (variant == 4) { a *= 64.f; b *= -64.f; }      // all products negative, C = 0
            if (variant == 5) { a *= 64.f; b *= 64.f; }       // ONE instruction from C = 0, positive products
            hA[i] = (_Float16)a; hB[i] = (_Float16)b;
        }
        _Float16 *dA, *dB; float* dout; (void)hipMalloc(&dA, (size_t)steps * 64 * 8 * 2); (void)hipMalloc(&dB, (size_t)steps * 64 * 8 * 2); (void)hipMalloc(&dout, 64 * 16 * 4);
        (void)hipMemcpy(dA, hA, (size_t)steps * 64 * 8 * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, (size_t)steps * 64 * 8 * 2, hipMemcpyHostToDevice);
        const float C = variant == 2 || variant >= 4 ? 0.f : (variant == 3 ? -65536.f : 65536.f);
        hipLaunchKernelGGL(kmfma, dim3(1), dim3(64), 0, 0, dA, dB, C, steps, dout);
        float ho[1024]; (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        // exact: D[i][j] = C + sum_s sum_k A_s[i][k] B_s[k][j];  lane l holds A[i = l & 31][k = 8 (l >> 5) + q], B[k = 8 (l >> 5) + q][j = l & 31]
        double sum_err = 0, sum_abs = 0, ulp = 0; int cnt = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
                const int j = lane & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                double ex = C;
                for (int s = 0; s < steps; ++s)
                    for (int k = 0; k < 16; ++k) {
                        const int la = i + 32 * (k >> 3), lb = j + 32 * (k >> 3), q = k & 7;
                        ex += (double)(float)hA[((size_t)s * 64 + la) * 8 + q] * (double)(float)hB[((size_t)s * 64 + lb) * 8 + q];
                    }
                const double u = ldexp(1.0, (int)floor(log2(fabs(ex))) - 23);
                sum_err += (ho[lane * 16 + r] - ex) / u; sum_abs += fabs(ho[lane * 16 + r] - ex) / u; ++cnt; ulp = u;
            }
        printf("variant %d (%s): %d MFMA steps: mean error %+.3f ulp of the result, mean |error| %.3f ulp\n", variant,
               variant == 0 ? "C = 65536, positive products ~1" : variant == 1 ? "C = 65536, random-sign products" : variant == 2 ? "C = 0, positive products ~4096" : variant == 3 ? "C = -65536, random-sign products" : variant == 4 ? "C = 0, negative products ~-4096" : "C = 0, positive products, ONE instruction", steps, sum_err / cnt, sum_abs / cnt);
    }
    return 0;
}
