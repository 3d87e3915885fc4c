// K1 of the fp64 path at size: one v_mfma_f64_16x16x4_f64 pass per gradient (k_big_f64.hip describes the whole path; this file
// depends on pmx_common.h alone so that scratch/f64pass_bench.hip can time the kernel by itself).
#include "pmx_common.h"
#include <algorithm>

__device__ __forceinline__ double p64_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
typedef double v4d __attribute__((ext_vector_type(4)));

struct Pass64Args {
    const double* Y;
    int64_t ldY;
    const double* F;         // fixed factor: rowsF x K (the gradient is taken with respect to it)
    const double* W;         // swept factor: rowsW x K
    double* slab;            // [nsplit][rowsF][K]
    double* lossPart;        // [gridDim.x] sum of T^2 over this workgroup's blocks, or nullptr
    const DevStatus* status;
    int rowsF, rowsW, K;
    int nsplit, bps;         // splits of the sweep (-> slabs), 64-row blocks of W per split
    int store;               // 0: the loss alone (pmx_loglike)
};
// column k of row r of a staged block sits at k ^ b64_swz(r): both operand patterns -- 16 rows x 4 columns (GEMM1's A operand)
// and 4 rows x 16 columns (the second product's B operand) -- then touch every pair of banks once per half-wave
__device__ __forceinline__ int b64_swz(int row) { return ((row & 1) << 4) | (row & 14); }

template <int KP, bool TRANS>
__global__ __launch_bounds__(256) void k64_grad_pass(Pass64Args a) {
    constexpr int KS = KP / 4, KJ = KP / 16, NLD = KP / 4;     // contraction steps, output tiles along K, doubles staged per thread and block
    extern __shared__ __attribute__((aligned(16))) double wl[];   // [2][64][KP]
    __shared__ double lred[4];
    if (chain_halted(a.status)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int strip = blockIdx.x / a.nsplit, split = blockIdx.x - strip * a.nsplit;
    const int K = a.K;
    const int nblk = (a.rowsW + 63) / 64;
    const int b0 = split * a.bps, b1 = b0 + a.bps < nblk ? b0 + a.bps : nblk;
    const int fcol = strip * 64 + 16 * wv + l15;               // this lane's row of F (GEMM1's B operand) == its column of T
    const bool fok = fcol < a.rowsF;
    double ff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + q;
        ff[s] = (fok && k < K) ? a.F[(int64_t)fcol * K + k] : 0.0;
    }
    v4d gacc[KJ];
#pragma unroll
    for (int kj = 0; kj < KJ; ++kj) gacc[kj] = (v4d){0.0, 0.0, 0.0, 0.0};
    double loss = 0.0;
    double wreg[NLD];
    v4d yv[4];
    auto w_load = [&](int b) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i, r = e / KP, k = e - r * KP;
            const int64_t w = (int64_t)b * 64 + r;
            wreg[i] = (w < a.rowsW && k < K) ? a.W[w * K + k] : 0.0;
        }
    };
    auto w_store = [&](double* buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i, r = e / KP, k = e - r * KP;
            buf[r * KP + (k ^ b64_swz(r))] = wreg[i];
        }
    };
    auto y_load = [&](int b) {       // T's accumulator layout: register r of tile mi = row 16 mi + q + 4 r, column l15
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t w = (int64_t)b * 64 + 16 * mi + q + 4 * r;
                const bool ok = fok && w < a.rowsW;
                yv[mi][r] = ok ? (TRANS ? a.Y[(int64_t)fcol * a.ldY + w] : a.Y[w * a.ldY + fcol]) : 0.0;
            }
    };
    if (b0 < b1) {
        w_load(b0);
        y_load(b0);
        w_store(wl);
    }
    __syncthreads();
    const int ga = b64_swz(l15);
    for (int b = b0; b < b1; ++b) {
        const double* cur = wl + ((b - b0) & 1) * 64 * KP;
        v4d t[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) t[mi] = -yv[mi];
        const bool more = b + 1 < b1;
        if (more) {                  // the next block's rows of W and tile of Y: in flight under this block's MFMAs
            w_load(b + 1);
            y_load(b + 1);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const double aop = cur[(16 * mi + l15) * KP + ((4 * s + q) ^ ga)];
                t[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, ff[s], t[mi], 0, 0, 0);
            }
        if (a.lossPart != nullptr) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) loss += t[mi][r] * t[mi][r];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mi + 4 * r + q, gb = b64_swz(4 * r + q);
#pragma unroll
                for (int kj = 0; kj < KJ; ++kj) {
                    const double bop = cur[row * KP + ((16 * kj + l15) ^ gb)];
                    gacc[kj] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[mi][r], bop, gacc[kj], 0, 0, 0);
                }
            }
        if (more) w_store(wl + (((b - b0) & 1) ^ 1) * 64 * KP);
        __syncthreads();
    }
    if (a.store) {
#pragma unroll
        for (int kj = 0; kj < KJ; ++kj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = strip * 64 + 16 * wv + q + 4 * r, kc = 16 * kj + l15;
                if (f < a.rowsF && kc < K) a.slab[((int64_t)split * a.rowsF + f) * K + kc] = gacc[kj][r];
            }
    }
    if (a.lossPart != nullptr) {
        loss = p64_wave_sum(loss);
        if (lane == 0) lred[wv] = loss;
        __syncthreads();
        if (tid == 0) a.lossPart[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
    }
}
// splits of a sweep over rowsW rows for a fixed factor of rowsF rows: enough workgroups for four per CU, slabs that stay a
// small fraction of Y's bytes (nsplit K / rowsW <= ~1/10), no empty split
inline void pass64_plan(int64_t rowsF, int64_t rowsW, int K, int* nsplit, int* bps) {
    const int strips = (int)((rowsF + 63) / 64), nblk = (int)((rowsW + 63) / 64);
    int ns = (1024 + strips - 1) / strips;
    const int cap = (int)std::max<int64_t>(1, rowsW / (10 * (int64_t)std::max(K, 1)));
    ns = std::min(std::min(ns, 32), std::min(cap, nblk));
    ns = std::max(ns, 1);
    const int per = (nblk + ns - 1) / ns;
    *bps = per;
    *nsplit = (nblk + per - 1) / per;
}
hipError_t launch_grad64_pass(const Pass64Args& a, int KP, bool trans, hipStream_t s) {
    const int strips = (a.rowsF + 63) / 64;
    const size_t lds = (size_t)2 * 64 * KP * sizeof(double);
    const dim3 grid(strips * a.nsplit), block(256);
#define PMX_PASS64(KPV, TR)                                                                                                              \
    do {                                                                                                                                 \
        hipError_t e_ = hipFuncSetAttribute((const void*)k64_grad_pass<KPV, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        if (e_ != hipSuccess) return e_;                                                                                                 \
        hipLaunchKernelGGL((k64_grad_pass<KPV, TR>), grid, block, lds, s, a);                                                            \
    } while (0)
    if (KP == 32) { if (trans) PMX_PASS64(32, true); else PMX_PASS64(32, false); }
    else if (KP == 64) { if (trans) PMX_PASS64(64, true); else PMX_PASS64(64, false); }
    else { if (trans) PMX_PASS64(128, true); else PMX_PASS64(128, false); }
#undef PMX_PASS64
    return hipGetLastError();
}

