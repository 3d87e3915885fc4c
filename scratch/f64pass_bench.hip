// [r6] times k64_grad_pass (proxmin_amd/csrc/k_grad_f64.hip) by itself: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I proxmin_amd/csrc scratch/f64pass_bench.hip -o scratch/f64pass_bench
// usage: f64pass_bench M N K [reps]; prints ms per pass (gSt, gA), TFLOP/s, and a checksum of each slab fold against a CPU sample
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include "pmx.h"
#include "k_grad_f64.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 16384, N = argc > 2 ? atoll(argv[2]) : 16384; const int K = argc > 3 ? atoi(argv[3]) : 64;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int KP = K <= 32 ? 32 : (K <= 64 ? 64 : 128);
    std::vector<double> hA(M * K), hS(N * K), hY(M * N);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hS) v = rnd() / K;
    for (auto& v : hY) v = rnd();
    double *A0, *S0, *A, *S, *Y, *slab[2], *loss; DevStatus* st;
    const int64_t Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64;
    CK(hipMalloc(&A0, M * K * 8)); CK(hipMalloc(&S0, N * K * 8)); CK(hipMalloc(&A, Mp * KP * 8)); CK(hipMalloc(&S, Np * KP * 8)); CK(hipMalloc(&Y, Mp * Np * 8)); CK(hipMalloc(&st, sizeof(DevStatus)));
    CK(hipMemset(st, 0, sizeof(DevStatus))); CK(hipMemset(Y, 0, Mp * Np * 8));
    CK(hipMemcpy(A0, hA.data(), M * K * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(S0, hS.data(), N * K * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy2D(Y, Np * 8, hY.data(), N * 8, N * 8, M, hipMemcpyHostToDevice));
    { Pad64Args pa{}; pa.X[0] = A0; pa.X[1] = S0; pa.P[0] = A; pa.P[1] = S; pa.rows[0] = M; pa.rows[1] = N; pa.K = K; pa.KP = KP; pa.status = st; launch_pad64(pa, nullptr); CK(hipDeviceSynchronize()); }
    int ns[2], bps[2];
    pass64_plan(M, N, K, &ns[0], &bps[0]); pass64_plan(N, M, K, &ns[1], &bps[1]);
    const int64_t rows[2] = {M, N};
    for (int j = 0; j < 2; ++j) CK(hipMalloc(&slab[j], (size_t)ns[j] * rows[j] * K * 8));
    CK(hipMalloc(&loss, 8 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int j = 1; j >= 0; --j) {
        Pass64Args p{};
        p.Y = Y; p.ldY = Np; p.F = j ? S : A; p.W = j ? A : S; p.rowsF = (int)rows[j]; p.rowsW = (int)rows[1 - j]; p.K = K;
        p.slab = slab[j]; p.status = st; p.nsplit = ns[j]; p.bps = bps[j]; p.store = 1; p.lossPart = j ? loss : nullptr;
        CK(launch_grad64_pass(p, KP, j == 0, nullptr));
        CK(hipDeviceSynchronize());
        float best = 1e30f, tot = 0.f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, nullptr));
            CK(launch_grad64_pass(p, KP, j == 0, nullptr));
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best; tot += t;
        }
        // check 4 entries against the CPU
        std::vector<double> h((size_t)ns[j] * rows[j] * K);
        CK(hipMemcpy(h.data(), slab[j], h.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int c = 0; c < 4; ++c) {
            const int64_t f = (c * 7919 + 13) % rows[j]; const int k = (c * 31 + 5) % K;
            double want = 0.0, scale = 0.0;
            for (int64_t w = 0; w < rows[1 - j]; ++w) {
                const int64_t m = j ? w : f, n = j ? f : w;
                double r = -hY[m * N + n];
                for (int kk = 0; kk < K; ++kk) r += hA[m * K + kk] * hS[n * K + kk];
                const double o = j ? hA[m * K + k] : hS[n * K + k];
                want += r * o; scale += fabs(r * o);
            }
            double got = 0.0;
            for (int q = 0; q < ns[j]; ++q) got += h[((size_t)q * rows[j] + f) * K + k];
            worst = fmax(worst, fabs(got - want) / scale);
        }
        const double fl = 4.0 * M * N * KP;
        printf("%s pass %lldx%lldx%d (KP %d) nsplit %d bps %d: best %.3f ms mean %.3f ms | %.1f TFLOP/s (padded K) | Y stream %.2f TB/s | rel err %.1e\n", j ? "gSt" : "gA ", (long long)M, (long long)N, K, KP,
               ns[j], bps[j], best, tot / reps, fl / (best * 1e-3) / 1e12, M * N * 8.0 / (best * 1e-3) / 1e12, worst);
    }
    return 0;
}
