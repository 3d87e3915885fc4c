// K1: fused residual-gradient kernel (the O(MNK) part of the nmf() hot path).
//
// Restates proxmin/nmf.py:39-41 (grad_likelihood, W=1) and nmf.py:25 (log_likelihood):
//     R = A S - Y ;  gA = R S^T ;  gS = A^T R ;  loss = 1/2 sum R^2
// with ONE pass over Y.  R never leaves the chip: each workgroup computes a 128 x BN tile of A S on
// the matrix cores, subtracts the Y tile it streams from HBM, parks R in LDS, and feeds it straight
// back into the two gradient contractions.  S is held transposed (St, N x K), so both outputs are
// "tall" (rows x K) and the kernel is symmetric in A and St.
//
// F32 mode: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 157 TFLOP/s peak on MI355X).
//
// Work decomposition (8 wavefronts = 512 threads per workgroup, one workgroup per CU):
//   region  = RP row panels (128 rows each, run-time) x CB column blocks (BN columns each)
//   step    = one 128 x BN block of Y:
//       GEMM1  P = A_panel (128 x KP) . St_blk^T (KP x BN)        -> 32x32 tiles, accumulators
//              R = P - Y                                          -> LDS  Rl[128][BN+1]
//       GEMM2  gA_panel (128 x KP) += R (128 x BN) . St_blk (BN x KP)   accumulates over the CB blocks
//       GEMM3  gSt_blk (BN x KP)   += R^T (BN x 128) . A_panel (128 x KP) accumulates over the RP panels
//   gA accumulators are flushed once per row panel into slab (column-region) of `slabA`,
//   gSt accumulators once per workgroup into slab (row-region) of `slabS`; the update kernels sum the
//   slabs in a fixed order (deterministic, no float atomics).
//
// LDS images are row-major with an odd leading dimension (KP+1 / BN+1) so that every ds_read_b32 /
// ds_write_b32 pattern used below (lanes walk rows, or lanes walk columns) is bank-conflict free.
#include <type_traits>
#include "pmx_common.h"

struct GradArgs {
    const float* Y;      // M x N (ldY)
    int64_t ldY;
    const float* W;      // M x N weights (ldW) or nullptr for W == 1 (nmf.py:13-41)
    int64_t ldW;
    const float* A;      // M x K
    const float* St;     // N x K
    float* slabA;        // [nSlabA][M][K]
    float* slabS;        // [nSlabS][N][K]
    double* lossPart;    // [gridDim.x * gridDim.y]
    const DevStatus* status;
    int M, N, K;
    int RP;              // row panels per workgroup
    int doA, doS;        // which gradients are wanted (bsdmm needs one at a time, nmf.py:181-185)
    // k_grad_f32_pc<.., CHAIN> only: gA accumulated in place along chains of workgroups (see GradV4Args in k_grad_bf16.hip)
    int chainL;
    unsigned* chainFlags;
    unsigned chainBase;
    DevStatus* wstatus;
    int chainInject;
    K1GramFold fold;     // [r6] k_grad_f32_pc: the step rule's Gram fold riding in the first workgroups (pmx_common.h)
};

template <int KP> struct GradCfg;
template <> struct GradCfg<32>  { static constexpr int BN = 128, G1T = 2, G2T = 1, G2SPLIT = 2, G3T = 1, G3SPLIT = 2; };
template <> struct GradCfg<64>  { static constexpr int BN = 128, G1T = 2, G2T = 1, G2SPLIT = 1, G3T = 1, G3SPLIT = 1; };
template <> struct GradCfg<128> { static constexpr int BN = 64,  G1T = 1, G2T = 2, G2SPLIT = 1, G3T = 1, G3SPLIT = 1; };

constexpr int GRAD_BM = 128;
constexpr int GRAD_CB = 4;
constexpr int GRAD_THREADS = 512;

// C/D layout of a 32x32 MFMA tile: register i of lane l holds (row, col) =
// ((i&3) + 8*(i>>2) + 4*(l>>5), l&31)
__device__ __forceinline__ int tile_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// HASW: weighted likelihood (a.W != nullptr).  A template switch, not a run-time branch: the per-element weight
// addresses in the unweighted instantiation cost it ~100 spilled VGPRs (K1 1.03 -> 1.43 ms at 16384 x 16384 x 64)
template <int KP, bool HASW>
__global__ __launch_bounds__(GRAD_THREADS, 2) void k_grad_f32(GradArgs a) {
    using C = GradCfg<KP>;
    constexpr int BN = C::BN;
    constexpr int LDK = KP + 1;   // Al / Sl leading dimension
    constexpr int LDR = BN + 1;   // Rl leading dimension
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Al = lds;                       // [128][LDK]
    float* Sl = Al + GRAD_BM * LDK;        // [BN][LDK]
    float* Rl = Sl + BN * LDK;             // [128][LDR]

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;          // wave 0..7
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    const int M = a.M, N = a.N, K = a.K;
    const int rowRegion = blockIdx.x, colRegion = blockIdx.y;
    const int row0 = rowRegion * a.RP * GRAD_BM;
    const int col0 = colRegion * GRAD_CB * BN;

    // ---- per-wave tile assignment --------------------------------------------------------------
    const int g1_mt = w >> 1;                                      // GEMM1 row tile (all KP)
    const int g1_nt0 = (C::G1T == 2) ? 2 * (w & 1) : (w & 1);       // first column tile
    const int g2_mt = w >> 1;                                      // GEMM2 output row tile
    const int g2_kt0 = (KP == 128) ? 2 * (w & 1) : ((KP == 64) ? (w & 1) : 0);
    const int g2_half = (C::G2SPLIT == 2) ? (w & 1) : 0;           // which half of the inner n range
    int g3_nt, g3_kt, g3_half;
    if (KP == 128) { g3_nt = w >> 2; g3_kt = w & 3; g3_half = 0; }
    else if (KP == 64) { g3_nt = w >> 1; g3_kt = w & 1; g3_half = 0; }
    else { g3_nt = w >> 1; g3_kt = 0; g3_half = w & 1; }
    constexpr int G2_INNER = BN / C::G2SPLIT;        // inner n length per wave
    constexpr int G3_INNER = GRAD_BM / C::G3SPLIT;   // inner m length per wave

    f32x16 accS[GRAD_CB];
#pragma unroll
    for (int cb = 0; cb < GRAD_CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accS[cb][i] = 0.f;
    f32x16 accA[C::G2T];

    float lossAcc = 0.f;

    // valid extent of this workgroup's region (uniform)
    int nrp = (M - row0 + GRAD_BM - 1) / GRAD_BM;
    if (nrp > a.RP) nrp = a.RP;
    int ncb = (N - col0 + BN - 1) / BN;
    if (ncb > GRAD_CB) ncb = GRAD_CB;
    const int nsteps = nrp * ncb;

    // The P accumulators double as the landing zone of the Y tile: Y for step s+1 is requested right
    // after step s has parked its residual in LDS and flies during GEMM2/GEMM3 of step s; GEMM1 then
    // accumulates A S on top of -Y, so R = A S - Y needs no extra registers and no subtraction pass.
    // Out-of-range rows/columns yield 0 and meet zero-padded operands, so their residual is exactly 0.
    f32x16 p[C::G1T];
    // per-lane element offset inside a 128 x BN block (32-bit); the block origin and the per-register
    // row offsets are wave-uniform and stay in SGPRs
    const int laneRow = g1_mt * 32 + 4 * hi;
    const int laneCol = g1_nt0 * 32 + l31;
    const int laneOff = laneRow * (int)a.ldY + laneCol;
    auto request_Y = [&](int prow0, int bcol0) {
        const float* blk = a.Y + (int64_t)prow0 * a.ldY + bcol0;
        if (prow0 + GRAD_BM <= M && bcol0 + BN <= N) {   // interior block (uniform branch)
#pragma unroll
            for (int t = 0; t < C::G1T; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float* rowp = blk + (int64_t)((i & 3) + 8 * (i >> 2)) * a.ldY + t * 32;
                    p[t][i] = __builtin_nontemporal_load(rowp + laneOff);   // Y is read once: keep it out of L2 / MALL (the slabs live there)
                }
        } else {                                           // edge block: clamp the address, select 0
            const int rmax = M - 1 - prow0, cmax = N - 1 - bcol0;   // >= 0
#pragma unroll
            for (int t = 0; t < C::G1T; ++t) {
                const int lc = laneCol + t * 32;
                const int cc = lc < cmax ? lc : cmax;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int lr = laneRow + (i & 3) + 8 * (i >> 2);
                    const int rr = lr < rmax ? lr : rmax;
                    const float v = blk[(int64_t)rr * a.ldY + cc];
                    p[t][i] = (lr <= rmax && lc <= cmax) ? v : 0.f;
                }
            }
        }
    };
    auto flush_gA = [&](int prow0) {
        const int slab = colRegion * C::G2SPLIT + g2_half;
        float* dst = a.slabA + (int64_t)slab * M * K;
#pragma unroll
        for (int t = 0; t < C::G2T; ++t) {
            const int kk = (g2_kt0 + t) * 32 + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gr = prow0 + g2_mt * 32 + tile_row(i, lane);
                if (gr < M && kk < K) dst[(int64_t)gr * K + kk] = accA[t][i];
            }
        }
    };

    // whole blocks with K == KP are one contiguous run of rows * KP floats: all 16-byte loads issued back to back,
    // then scattered into the odd-stride LDS image.  (The guarded element loop below compiles to one dependent
    // 4-byte load + full wait per element: 16 serial L2 round trips per thread and step.)
    auto stage_rows_whole = [&](float* img, const float* src, auto rows_tag) {
        constexpr int ROWS = decltype(rows_tag)::value;
        constexpr int NV = ROWS * (KP / 4) / GRAD_THREADS;        // float4 per thread
        static_assert(ROWS * (KP / 4) % GRAD_THREADS == 0, "");
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = s4[tid + i * GRAD_THREADS];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * GRAD_THREADS;
            const int r = f / (KP / 4), k = (f - r * (KP / 4)) * 4;
            float* d = img + r * LDK + k;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
    };
    if (nsteps > 0) request_Y(row0, col0);
    int rp = 0, cb = 0;
#pragma nounroll
    for (int step = 0; step < nsteps; ++step) {
        const int prow0 = row0 + rp * GRAD_BM;
        const int bcol0 = col0 + cb * BN;
        __syncthreads();   // previous step's readers of Al / Sl / Rl are done
        if (cb == 0) {     // new row panel (uniform)
            // ---- stage the A panel: Al[m][k] = A[prow0+m][k], zero padded -----------------------
            if (K == KP && prow0 + GRAD_BM <= M) stage_rows_whole(Al, a.A + (int64_t)prow0 * K, std::integral_constant<int, GRAD_BM>{});
            else
#pragma unroll 4
                for (int e = tid; e < GRAD_BM * KP; e += GRAD_THREADS) {
                    const int m = e / KP, k = e - m * KP;
                    float v = 0.f;
                    if (prow0 + m < M && k < K) v = a.A[(int64_t)(prow0 + m) * K + k];
                    Al[m * LDK + k] = v;
                }
#pragma unroll
            for (int t = 0; t < C::G2T; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) accA[t][i] = 0.f;
        }
        // ---- stage the St block: Sl[n][k] = St[bcol0+n][k] --------------------------------------
        if (K == KP && bcol0 + BN <= N) stage_rows_whole(Sl, a.St + (int64_t)bcol0 * K, std::integral_constant<int, BN>{});
        else
#pragma unroll 4
            for (int e = tid; e < BN * KP; e += GRAD_THREADS) {
                const int n = e / KP, k = e - n * KP;
                float v = 0.f;
                if (bcol0 + n < N && k < K) v = a.St[(int64_t)(bcol0 + n) * K + k];
                Sl[n * LDK + k] = v;
            }
        __syncthreads();
        // ---- GEMM1: P = A S accumulated on top of -Y ---------------------------------------------
#pragma unroll
        for (int t = 0; t < C::G1T; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) p[t][i] = -p[t][i];
        {
            const float* ap = Al + (g1_mt * 32 + l31) * LDK + hi * (KP / 2);
            const float* bp = Sl + (g1_nt0 * 32 + l31) * LDK + hi * (KP / 2);
#pragma unroll 8
            for (int s = 0; s < KP / 2; ++s) {
                const float av = ap[s];
#pragma unroll
                for (int t = 0; t < C::G1T; ++t) {
                    const float bv = bp[t * 32 * LDK + s];
                    p[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, p[t], 0, 0, 0);
                }
            }
        }
        // ---- loss, park R in LDS ---------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < C::G1T; ++t) {
            const int lc = (g1_nt0 + t) * 32 + l31;
            // weights: ONE 64-bit row pointer per tile and 32-bit offsets within it (sixteen 64-bit addresses spill)
            const int gr0 = prow0 + g1_mt * 32 + 4 * hi, gc = bcol0 + lc;
            const float* wrow = nullptr;
            if constexpr (HASW) wrow = a.W + (int64_t)gr0 * a.ldW + gc;
            const int ldw = HASW ? (int)a.ldW : 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lr = g1_mt * 32 + tile_row(i, lane);
                float r = p[t][i];
                if constexpr (HASW) {        // weighted likelihood: loss 1/2 sum W d^2, D = W d
                    const int dr = (i & 3) + 8 * (i >> 2);           // tile_row(i, lane) - 4 * hi
                    const float wv = (gr0 + dr < M && gc < N) ? wrow[dr * ldw] : 0.f;
                    lossAcc += wv * (r * r);
                    r *= wv;
                } else {
                    lossAcc += r * r;
                }
                Rl[lr * LDR + lc] = r;
            }
        }
        __syncthreads();
        // ---- request the next step's Y tile (lands during GEMM2/GEMM3) ---------------------------
        int nrp_ = rp, ncb_ = cb + 1;
        if (ncb_ == ncb) { ncb_ = 0; nrp_ = rp + 1; }
        if (step + 1 < nsteps) request_Y(row0 + nrp_ * GRAD_BM, col0 + ncb_ * BN);

        // ---- GEMM2: gA(rows of this panel) += R . St_blk ----------------------------------------
        if (a.doA) {
            const float* rq = Rl + (g2_mt * 32 + l31) * LDR + g2_half * G2_INNER + hi * (G2_INNER / 2);
            const float* sp = Sl + (g2_half * G2_INNER + hi * (G2_INNER / 2)) * LDK + g2_kt0 * 32 + l31;
#pragma unroll 8
            for (int s = 0; s < G2_INNER / 2; ++s) {
                const float av = rq[s];
#pragma unroll
                for (int t = 0; t < C::G2T; ++t) {
                    const float bv = sp[s * LDK + t * 32];
                    accA[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accA[t], 0, 0, 0);
                }
            }
        }
        // ---- GEMM3: gSt(rows of this column block) += R^T . A_panel -----------------------------
        if (a.doS) {
            const float* rq = Rl + (g3_half * G3_INNER + hi * (G3_INNER / 2)) * LDR + g3_nt * 32 + l31;
            const float* aq = Al + (g3_half * G3_INNER + hi * (G3_INNER / 2)) * LDK + g3_kt * 32 + l31;
            // the accumulator of column block cb must be addressed statically to stay in registers:
            // uniform switch around the (small) MFMA loop instead of unrolling the whole step
#define GEMM3_INTO(ACC)                                                                       \
    _Pragma("unroll 8") for (int s = 0; s < G3_INNER / 2; ++s)                                \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(rq[s * LDR], aq[s * LDK], ACC, 0, 0, 0);
            switch (cb) {
                case 0: GEMM3_INTO(accS[0]) break;
                case 1: GEMM3_INTO(accS[1]) break;
                case 2: GEMM3_INTO(accS[2]) break;
                default: GEMM3_INTO(accS[3]) break;
            }
#undef GEMM3_INTO
        }
        // ---- end of a row panel: flush gA -------------------------------------------------------
        if (cb + 1 == ncb) {
            if (a.doA) flush_gA(prow0);
        }
        cb = ncb_;
        rp = nrp_;
    }
    // ---- flush gSt: slab = rowRegion (x split) ------------------------------------------------------
    if (a.doS) {
        const int slab = rowRegion * C::G3SPLIT + g3_half;
        float* dst = a.slabS + (int64_t)slab * N * K;
        const int kk = g3_kt * 32 + l31;
#pragma unroll
        for (int cb = 0; cb < GRAD_CB; ++cb) {
            const int bcol0 = col0 + cb * BN;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gn = bcol0 + g3_nt * 32 + tile_row(i, lane);
                if (gn < N && kk < K) dst[(int64_t)gn * K + kk] = accS[cb][i];
            }
        }
    }
    // ---- loss partial (one double per workgroup) ---------------------------------------------------
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if (lane == 0) lds[w] = v;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < GRAD_THREADS / 64; ++i) s += (double)lds[i];
            a.lossPart[blockIdx.y * gridDim.x + blockIdx.x] = s;
        }
    }
}

// host-side launcher -----------------------------------------------------------------------------
struct GradPlan {
    int KP, BN, RP, gridX, gridY, nSlabA, nSlabS;
    size_t ldsBytes;
    int variant;   // split-bf16 kernels only: which K1 implementation whole-block shapes take (PMX_K1_VARIANT, tuning A/B)
};

// Row regions of a grid whose workgroups sit one (or `slots` / 256) per CU: `panels` row panels, gridY column regions.  A region of RP
// panels costs RP panel-times and the grid runs in ceil(gridX gridY / slots) rounds, so take the gridX with the smallest
// rounds x RP -- the fewest row regions among equals (= gSt slabs).  (Rounds 1-3 took ceil(slots / gridY) row regions: exact for
// the aligned bench shapes, but e.g. 125 panels x 63 column regions became 5 x 63 = 315 workgroups, a second round for 59 of them.)
static void plan_row_regions(int64_t panels, int gridY, int slots, int* RP, int* gridX) {
    int64_t best = 1, bestCost = -1;
    const int64_t hi = panels < 4096 ? panels : 4096;
    for (int64_t x = 1; x <= hi; ++x) {
        const int64_t rp = (panels + x - 1) / x;
        const int64_t gx = (panels + rp - 1) / rp;              // the regions that many panels per region really make
        const int64_t rounds = (gx * gridY + slots - 1) / slots;
        const int64_t cost = rounds * rp;
        if (bestCost < 0 || cost < bestCost) { bestCost = cost; best = x; }
    }
    *RP = (int)((panels + best - 1) / best);
    *gridX = (int)((panels + *RP - 1) / *RP);
}

GradPlan grad_plan_f32(int64_t M, int64_t N, int64_t K) {
    GradPlan p{};
    p.KP = K <= 32 ? 32 : (K <= 64 ? 64 : 128);
    p.BN = p.KP == 128 ? 64 : 128;
    const int splitA = p.KP == 32 ? 2 : 1, splitS = p.KP == 32 ? 2 : 1;
    const int64_t panels = (M + GRAD_BM - 1) / GRAD_BM;
    p.gridY = (int)((N + (int64_t)GRAD_CB * p.BN - 1) / ((int64_t)GRAD_CB * p.BN));
    // aim for >= 2 workgroups per CU, but keep the number of gSt slabs (= row regions) small
    int64_t wantX = (512 + p.gridY - 1) / p.gridY;
    if (wantX < 1) wantX = 1;
    if (wantX > panels) wantX = panels;
    p.RP = (int)((panels + wantX - 1) / wantX);
    p.gridX = (int)((panels + p.RP - 1) / p.RP);
    p.nSlabA = p.gridY * splitA;
    p.nSlabS = p.gridX * splitS;
    p.ldsBytes = sizeof(float) * ((size_t)GRAD_BM * (p.KP + 1) + (size_t)p.BN * (p.KP + 1) + (size_t)GRAD_BM * (p.BN + 1));
    return p;
}

// ------------------------------------------------------------------------------------------------
// k_grad_small: the fused residual-gradient pass for SMALL problems (K <= 16, a few million entries of Y -- the
// reference's own examples are 100 x 50 x 3 and 200 x 1000 x 5).  The matrix-core kernels above are all latency at that
// size (k_grad_f32<32> on a ragged 200 x 1000 x 5: 54 us of staging, barriers and guarded loads for 6 MFLOP); here a
// workgroup of 256 threads owns a 16 x 256 tile of Y, thread = column:
//   phase 1  r[m] = A[m,:] . S[:,n] - Y[m,n] for the tile's 16 rows (A tile broadcast from LDS, S column in registers),
//            gS[:,n] += A[m,:] r  in registers, r parked in LDS;
//   phase 2  gA[m,:] over the tile's 256 columns from the parked r and the S tile (16 lanes per (row, component)).
// Plain fp32 FMAs in a fixed order; same outputs as the other K1s: one gSt slab per row tile, one gA slab per column
// tile (summed in a fixed order by the update kernels), a loss partial per workgroup.  Weights W as a run-time branch.
// ------------------------------------------------------------------------------------------------
constexpr int SG_ROWS = 16, SG_COLS = 256, SG_KMAX = 16;
// shared memory of one tile, carved out of a caller-provided pool (k_small_front shares the pool with the step-rule role)
template <int KM>
struct SmallTileSmem {
    float As[SG_ROWS][KM + 1];
    float Ss[KM][SG_COLS + 1];
    float Rs[SG_ROWS][SG_COLS + 1];
    float lred[SG_COLS / 64];
};
// tile (bx, by) of a (gx column tiles) x (row tiles) grid; KM = 8 or 16 >= K: the loops over components are unrolled to KM
template <int KM>
__device__ __forceinline__ void grad_small_tile(const GradArgs& a, SmallTileSmem<KM>& sm, int bx, int by, int gx) {
    const int tid = threadIdx.x, K = a.K;
    const int row0 = by * SG_ROWS, col0 = bx * SG_COLS;
    const int n = col0 + tid;
    const bool nval = n < a.N;
    // the halt flag is requested together with the operands and looked at when they are there: at this size a dependent
    // round trip to memory (~1.3 us) is a sixth of the kernel
    const int halted = __builtin_nontemporal_load(&a.status->halt);
    float sv[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        sv[k] = (nval && k < K) ? a.St[(int64_t)n * K + k] : 0.f;
        sm.Ss[k][tid] = sv[k];
    }
    for (int e = tid; e < SG_ROWS * KM; e += SG_COLS) {
        const int r = e / KM, k = e - r * KM;
        sm.As[r][k] = (row0 + r < a.M && k < K) ? a.A[(int64_t)(row0 + r) * K + k] : 0.f;
    }
    float yv[SG_ROWS], wv[SG_ROWS];
    const bool hasw = a.W != nullptr;
#pragma unroll
    for (int r = 0; r < SG_ROWS; ++r) {
        const bool ok = nval && row0 + r < a.M;
        yv[r] = ok ? a.Y[(int64_t)(row0 + r) * a.ldY + n] : 0.f;
        wv[r] = (ok && hasw) ? a.W[(int64_t)(row0 + r) * a.ldW + n] : 1.f;
    }
    if (halted) return;                      // (uniform; nothing has been written yet)
    __syncthreads();
    float gs[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) gs[k] = 0.f;
    float loss = 0.f;
#pragma unroll
    for (int r = 0; r < SG_ROWS; ++r) {
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) p += sm.As[r][k] * sv[k];
        float rr = (nval && row0 + r < a.M) ? p - yv[r] : 0.f;
        loss += wv[r] * (rr * rr);
        rr *= wv[r];
        sm.Rs[r][tid] = rr;
#pragma unroll
        for (int k = 0; k < KM; ++k) gs[k] += sm.As[r][k] * rr;
    }
    if (a.doS && nval) {
        float* dst = a.slabS + ((int64_t)by * a.N + n) * K;
        for (int k = 0; k < K; ++k) dst[k] = gs[k];
    }
    __syncthreads();
    if (a.doA & 1) {
        // gA[m, k] = sum over the tile's 256 columns of r[m, c] S[k, c]: 16 lanes per output (columns sub, sub + 16, ..),
        // 16 outputs per pass of the workgroup, fixed order within a lane and across the 16 lanes
        const int sub = tid & 15, og = tid >> 4;
        for (int o0 = 0; o0 < SG_ROWS * K; o0 += SG_COLS / 16) {
            const int o = o0 + og;
            const bool oval = o < SG_ROWS * K;
            const int r = oval ? o / K : 0, k = oval ? o - r * K : 0;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < SG_COLS / 16; ++c) s += sm.Rs[r][sub + 16 * c] * sm.Ss[k][sub + 16 * c];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (oval && sub == 0 && row0 + r < a.M) a.slabA[((int64_t)bx * a.M + row0 + r) * K + k] = s;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
    if ((tid & 63) == 0) sm.lred[tid >> 6] = loss;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < SG_COLS / 64; ++i) s += (double)sm.lred[i];
        a.lossPart[by * gx + bx] = s;
    }
}
template <int KM>
__global__ __launch_bounds__(SG_COLS) void k_grad_small(GradArgs a) {
    __shared__ SmallTileSmem<KM> sm;
    grad_small_tile<KM>(a, sm, blockIdx.x, blockIdx.y, gridDim.x);
}
// the small-problem kernel takes: K <= 16 and a Y of at most 1 M entries (at 2000 x 2000 x 16 the matrix-core kernels are
// ahead again: 163 against 183 us per pgm iteration)
bool grad_small_applies(int64_t M, int64_t N, int64_t K) {
    if (getenv("PMX_K1_SMALL") && atoi(getenv("PMX_K1_SMALL")) == 0) return false;
    return K <= SG_KMAX && M <= 4096 && N <= 16384 && M * N <= ((int64_t)1 << 20);
}
GradPlan grad_plan_small(int64_t M, int64_t N, int64_t K) {
    GradPlan p{};
    p.KP = 32;
    p.BN = SG_COLS;
    p.RP = 1;
    p.gridX = (int)((M + SG_ROWS - 1) / SG_ROWS);       // row tiles   (-> gSt slabs)
    p.gridY = (int)((N + SG_COLS - 1) / SG_COLS);       // column tiles (-> gA slabs)
    p.nSlabA = p.gridY;
    p.nSlabS = p.gridX;
    p.ldsBytes = 0;
    p.variant = -1;
    (void)K;
    return p;
}
hipError_t grad_launch_small(const GradPlan& p, const GradArgs& a, hipStream_t stream) {
    if (a.K <= 8) hipLaunchKernelGGL(k_grad_small<8>, dim3(p.gridY, p.gridX), dim3(SG_COLS), 0, stream, a);
    else hipLaunchKernelGGL(k_grad_small<16>, dim3(p.gridY, p.gridX), dim3(SG_COLS), 0, stream, a);
    return hipGetLastError();
}

template <int KP, bool HASW>
static hipError_t grad_launch_f32_t(const GradPlan& p, const GradArgs& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f32<KP, HASW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.ldsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f32<KP, HASW>), dim3(p.gridX, p.gridY), dim3(GRAD_THREADS), p.ldsBytes, stream, a);
    return hipGetLastError();
}
hipError_t grad_launch_f32(const GradPlan& p, const GradArgs& a, hipStream_t stream) {
    const bool w = a.W != nullptr;
    switch (p.KP) {
        case 32: return w ? grad_launch_f32_t<32, true>(p, a, stream) : grad_launch_f32_t<32, false>(p, a, stream);
        case 64: return w ? grad_launch_f32_t<64, true>(p, a, stream) : grad_launch_f32_t<64, false>(p, a, stream);
        default: return w ? grad_launch_f32_t<128, true>(p, a, stream) : grad_launch_f32_t<128, false>(p, a, stream);
    }
}
