// K1 of the fp64 path at size: one v_mfma_f64_16x16x4_f64 pass per gradient (k_big_f64.hip describes the whole path; this file
// depends on pmx_common.h alone so that scratch/f64pass_bench.hip can time the kernel by itself).
#include "pmx_common.h"
#include <algorithm>
#include <type_traits>

__device__ __forceinline__ double p64_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

struct Pass64Args {
    const double* Y;
    int64_t ldY;
    const double* Wt;        // weights of the likelihood (nmf.py:13-41), Y's shape and pitch; nullptr: W == 1 (the <HASW = false> instances)
    const double* F;         // fixed factor: rowsF x K (the gradient is taken with respect to it)
    const double* W;         // swept factor: rowsW x K
    double* slab;            // [nsplit][rowsF][K]
    double* lossPart;        // [gridDim.x] sum of T^2 over this workgroup's blocks, or nullptr
    const DevStatus* status;
    int rowsF, rowsW, K;     // the REAL extents (slab rows, components stored)
    int nsplit, bps;         // splits of the sweep (-> slabs), 64-row blocks of W per split
    int store;               // 0: the loss alone (pmx_loglike)
};
// column k of row r of a staged block sits at k ^ b64_swz(r): both operand patterns -- 16 rows x 4 columns (GEMM1's A operand)
// and 4 rows x 16 columns (the second product's B operand) -- then touch every pair of banks once per half-wave
__device__ __forceinline__ int b64_swz(int row) { return ((row & 1) << 4) | (row & 14); }

template <int KP, bool TRANS, bool HASW>
__global__ __launch_bounds__(256, (KP <= 64 && !HASW ? 2 : 1)) void k64_grad_pass(Pass64Args a) {
    constexpr int KS = KP / 4, KJ = KP / 16, NLD = KP / 4;     // contraction steps, output tiles along K, doubles staged per thread and block
    // A block is 2 KP MFMAs per wave, issued in chunks of CH; everything else of the step hangs on that sequence (the scheduler is held to
    // it: left alone it hoists every LDS read of the block and spills, and a burst of 32 global loads in front of the first MFMA keeps the
    // wave from issuing anything else for ~2000 cycles):
    //   chunk c          requests the LDS operands of chunk c + 1 (two register sets), then issues its MFMAs
    //   chunks 0..NG-1   request the NEXT block's rows of W (L2-resident: every workgroup sweeps them), a group of GS per chunk, and store them
    //                    to the other LDS buffer WD chunks later
    //   chunk N1-1       GEMM1 is complete: T = W F^T - Y (nmf.py:39) with the tile of Y requested during the PREVIOUS block
    //   chunks N1..      request the NEXT block's tile of Y (HBM: ~2 us away; it has the rest of this block and the next GEMM1 to arrive.
    //                    Requested at the head of its own block it was not back when GEMM1 ended: 0.2 ms of a 1.4 ms pass at cfg3)
    constexpr int CH = 8, N1 = 4 * KS / CH, NCH = (4 * KS + 16 * KJ) / CH;
    constexpr int YC = N1 >= 8 ? 4 : 2, YPC = 16 / YC;
    constexpr int GS = 4, NG = NLD / GS, WD = 3;
    static_assert((4 * KS) % CH == 0 && (16 * KJ) % CH == 0 && NG + WD <= NCH && N1 + YC <= NCH, "chunk plan");
    extern __shared__ __attribute__((aligned(16))) double wl[];   // [2][64][KP]
    __shared__ double lred[4];
    if (chain_halted(a.status)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int strip = blockIdx.x / a.nsplit, split = blockIdx.x - strip * a.nsplit;
    const int K = a.K;
    const int nblk = (a.rowsW + 63) / 64;       // (the padding makes the last one whole)
    const int b0 = split * a.bps, b1 = b0 + a.bps < nblk ? b0 + a.bps : nblk;
    const int fcol = strip * 64 + 16 * wv + l15;               // this lane's row of F (GEMM1's B operand) == its column of T
    // F, W and Y are PADDED by the caller: rows to multiples of 64, K to KP (row pitch KP), zeros behind the real extents (k64_pad_factors
    // below) -- no test on any load
    double ff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) ff[s] = a.F[(int64_t)fcol * KP + 8 * (s >> 1) + 2 * q + (s & 1)];      // contraction step s takes k = 8 (s >> 1) + 2 q + (s & 1): steps 2 c, 2 c + 1 are ONE 16-byte LDS read of W
    v4d gacc[KJ];
#pragma unroll
    for (int kj = 0; kj < KJ; ++kj) gacc[kj] = (v4d){0.0, 0.0, 0.0, 0.0};
    double loss = 0.0;
    const double* ylane = a.Y + (int64_t)fcol * a.ldY + q;      // TRANS: this lane's corner of a tile
    const int64_t yoff = (int64_t)q * a.ldY + fcol;             // else: its offset from a tile's first row
    v4d yv[4], wv4[HASW ? 4 : 1];
    const double* wlane = HASW ? a.Wt + (int64_t)fcol * a.ldY + q : nullptr;
    auto y_req = [&](int b, int idx) {      // T's accumulator layout: register r of tile mi = row 16 mi + q + 4 r, column l15
        const int mi = idx >> 2, r = idx & 3;
        if (TRANS) yv[mi][r] = ylane[(int64_t)b * 64 + 16 * mi + 4 * r];                 // one address per lane + immediates
        else yv[mi][r] = (a.Y + ((int64_t)b * 64 + 16 * mi + 4 * r) * a.ldY)[yoff];      // a scalar row base + one offset per lane
        if constexpr (HASW) {
            if (TRANS) wv4[mi][r] = wlane[(int64_t)b * 64 + 16 * mi + 4 * r];
            else wv4[mi][r] = (a.Wt + ((int64_t)b * 64 + 16 * mi + 4 * r) * a.ldY)[yoff];
        }
    };
    if (b0 < b1) {                   // the first block's rows of W and tile of Y
        const double* wb = a.W + (int64_t)b0 * 64 * KP;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i, r = e / KP, k = e - r * KP;
            wl[r * KP + (k ^ b64_swz(r))] = wb[e];
        }
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) y_req(b0, idx);
    }
    __syncthreads();
    const unsigned wS = (unsigned)(8 * ((tid / KP) * KP + ((tid % KP) ^ b64_swz(tid / KP))));       // this thread's slot of a staged block, rows tid / KP + (256 / KP) jj
    const unsigned laR = (unsigned)(8 * (l15 * KP + ((2 * q) ^ b64_swz(l15)))), lbR = (unsigned)(8 * (q * KP + (l15 ^ b64_swz(q))));
    for (int b = b0; b < b1; ++b) {
        const unsigned nxtB = (unsigned)((((b - b0) & 1) ^ 1) * 64 * KP * 8);
        const bool more = b + 1 < b1;
        const double* wb = a.W + (int64_t)(b + 1) * 64 * KP;
        v4d t[4];
        double op[2][CH], wtmp[NG][GS];
        // operand addresses (bytes): column k of row r sits at k ^ b64_swz(r).
        //   GEMM1's A operand: row 16 mi + l15, columns 8 c + 2 q + {0, 1}  ->  [l15 KP + (2 q ^ swz(l15))] ^ 8 (c & 3), + 32 (c >> 2) + 16 mi KP
        //   (swz is even: the pair stays adjacent and 16-byte aligned);
        //   the second product's B operand: row 16 mi + 4 r + q, column 16 kj + l15  ->  [q KP + (l15 ^ swz(q))] ^ (16 (kj & 1) ^ 4 r), + 32 (kj >> 1) +
        //   (16 mi + 4 r) KP   (swz(4 r + q) = swz(q) ^ 4 r).
        // The bracket is one lane constant per product (+ this block's buffer); what is XORed touches bits 2..4 of the column only, everything else is
        // an immediate offset of the read: one v_xor per variant and chunk (the asm keeps the compiler from hoisting all twelve variants out of the
        // chunks -- that spilled --, per-operand address arithmetic cost 3.8 VALU instructions per MFMA)
        const unsigned curB = (unsigned)(((b - b0) & 1) * 64 * KP * 8);
        const char* ldsb = reinterpret_cast<const char*>(wl);
        auto opl = [&](auto cc, int buf) {
            constexpr int c = decltype(cc)::value;
            unsigned la = laR + curB, lb = lbR + curB;
            asm volatile("" : "+v"(la), "+v"(lb));
            if constexpr (c < N1) {                  // chunk c of GEMM1 = contraction steps 2 c and 2 c + 1 of the four tiles: op[h * 4 + mi]
                const unsigned base = la ^ (unsigned)(64 * (c & 3));
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const v2d pr = *reinterpret_cast<const v2d*>(ldsb + base + (256 * (c >> 2) + 16 * mi * KP * 8));
                    op[buf][mi] = pr[0];
                    op[buf][4 + mi] = pr[1];
                }
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int h = (c - N1) * CH + i, kj = h % KJ, mr = h / KJ, mi = mr >> 2, r = mr & 3;
                    const unsigned base = lb ^ (unsigned)(8 * ((16 * (kj & 1)) ^ (4 * r)));
                    op[buf][i] = *reinterpret_cast<const double*>(ldsb + base + (256 * (kj >> 1) + (16 * mi + 4 * r) * KP * 8));
                }
            }
        };
        auto run = [&](auto self, auto cc) -> void {
            constexpr int c = decltype(cc)::value;
            if constexpr (c < NG) {
                if (more) {
#pragma unroll
                    for (int i = 0; i < GS; ++i) wtmp[c][i] = wb[tid + 256 * (c * GS + i)];
                }
            }
            if constexpr (c >= WD && c < WD + NG) {
                if (more) {
#pragma unroll
                    for (int i = 0; i < GS; ++i) {       // element tid + 256 jj of the block: row r0 + (256 / KP) jj, column tid % KP  ->  wS ^ 8 (rj & 14), + 8 rj KP
                        constexpr int rj = (256 / KP) * ((c - WD) * GS);
                        const int rji = rj + (256 / KP) * i;
                        *reinterpret_cast<double*>(reinterpret_cast<char*>(wl) + ((wS + nxtB) ^ (unsigned)(8 * (rji & 14))) + 8 * rji * KP) = wtmp[c - WD][i];
                    }
                }
            }
            if constexpr (c >= N1 && c < N1 + YC) {
                if (more) {
#pragma unroll
                    for (int i = 0; i < YPC; ++i) y_req(b + 1, (c - N1) * YPC + i);
                }
            }
            if constexpr (c + 1 < NCH) opl(std::integral_constant<int, c + 1>{}, (c + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if constexpr (c < N1) {
                    const int g = c * CH + i, s1 = g >> 2, mi = g & 3;
                    if constexpr (c == 0) {
                        if (i < 4) t[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[c & 1][i], ff[s1], (v4d){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);      // (the block's first product of a tile: C = 0 inline)
                        else t[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[c & 1][i], ff[s1], t[mi], 0, 0, 0);
                    } else t[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[c & 1][i], ff[s1], t[mi], 0, 0, 0);
                } else {
                    const int h = (c - N1) * CH + i, kj = h % KJ, mr = h / KJ, mi = mr >> 2, r = mr & 3;
                    gacc[kj] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[mi][r], op[c & 1][i], gacc[kj], 0, 0, 0);
                }
            }
            if constexpr (c == N1 - 1) {     // the residual D = W (A S - Y) (nmf.py:39), and sum W (A S - Y)^2 for the likelihood (nmf.py:25)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) t[mi] -= yv[mi];
                if (a.lossPart != nullptr) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) loss += HASW ? wv4[mi][r] * (t[mi][r] * t[mi][r]) : t[mi][r] * t[mi][r];
                }
                if constexpr (HASW) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) t[mi] *= wv4[mi];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (c + 1 < NCH) self(self, std::integral_constant<int, c + 1>{});
        };
        opl(std::integral_constant<int, 0>{}, 0);
        run(run, std::integral_constant<int, 0>{});
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
// a factor (rows x K, row-major) into its padded copy for K1: ceil64(rows) x KP, zeros behind the real extents
struct Pad64Args {
    const double* X[2];
    double* P[2];
    int64_t rows[2];
    int K, KP;
    const DevStatus* status;
};
__global__ __launch_bounds__(256) void k64_pad_factors(Pad64Args a) {
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int64_t rp = (a.rows[j] + 63) / 64 * 64, n = rp * a.KP;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / a.KP;
        const int k = (int)(e - r * a.KP);
        a.P[j][e] = (r < a.rows[j] && k < a.K) ? a.X[j][r * a.K + k] : 0.0;
    }
}
inline void launch_pad64(const Pad64Args& a, hipStream_t s) { hipLaunchKernelGGL(k64_pad_factors, dim3(512, 2), dim3(256), 0, s, a); }
inline bool pad64_needed(int64_t rows, int K, int KP) { return rows % 64 != 0 || K != KP; }

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
#define PMX_PASS64(KPV, TR, HW)                                                                                                              \
    do {                                                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute((const void*)k64_grad_pass<KPV, TR, HW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        if (e_ != hipSuccess) return e_;                                                                                                     \
        hipLaunchKernelGGL((k64_grad_pass<KPV, TR, HW>), grid, block, lds, s, a);                                                            \
    } while (0)
#define PMX_PASS64_T(KPV)                                                                                         \
    do {                                                                                                          \
        if (a.Wt != nullptr) { if (trans) PMX_PASS64(KPV, true, true); else PMX_PASS64(KPV, false, true); }       \
        else { if (trans) PMX_PASS64(KPV, true, false); else PMX_PASS64(KPV, false, false); }                     \
    } while (0)
    if (KP == 32) PMX_PASS64_T(32);
    else if (KP == 64) PMX_PASS64_T(64);
    else PMX_PASS64_T(128);
#undef PMX_PASS64_T
#undef PMX_PASS64
    return hipGetLastError();
}

