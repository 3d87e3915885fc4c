// K1, split-bf16 variant: the same fused residual-gradient pass as k_grad.hip
//     R = A S - Y ;  gA = R S^T ;  gS = A^T R ;  loss = 1/2 sum R^2        (proxmin/nmf.py:25,39-41)
// but on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) with fp32-class
// accuracy: every fp32 operand x is split into bf16 terms x = x0 + x1 (+ x2) and the products that
// matter are accumulated in fp32:
//     A S      : 3-term split of A and S, 6 products  a0s0 a0s1 a1s0 a0s2 a1s1 a2s0   (error ~2^-24)
//     R S^T    : 2-term split of R and S, 3 products  r0s0 r0s1 r1s0                   (error ~2^-17,
//     A^T R    : 2-term split of A and R, 3 products                                    no cancellation)
// The residual is where cancellation happens (|R| << |A S| near a solution), hence the third term
// there; measured gradient error vs fp64 equals the pure-fp32 path's ("split-bf16").
// 12 bf16 MFMA passes replace 3 fp32 passes = 4x fewer matrix-core cycles.
//
// k_presplit writes, once per gradient evaluation, the bf16 terms of both factors in the two
// orientations the contractions need (row-major with K contiguous, and transposed with the long
// dimension contiguous), zero-padded to KP columns and to a multiple of 128 rows, so the hot kernel
// stages them into LDS with unguarded 16-byte copies.
//
// Workgroup = 8 wavefronts, step = 128 x 64 block of Y (KP = 64 or 32):
//   GEMM1  wave w: P tile (mt = w>>1, nt = w&1); A terms live in registers for the whole row panel,
//          S terms from LDS (Sl, ds_read_b128); P accumulates on top of -Y (prefetched into the acc).
//   R      parked once in LDS as fp32 (Rl, stride 68: rows as ds_read_b128, columns as ds_read_b32,
//          both conflict-free); split into bf16 terms by whichever wave consumes it.
//   GEMM2  wave w: gA tile (mt = w>>1, kt = w&1), contraction over the block's 64 columns.
//   GEMM3  wave w: gSt tile (nt = w&1, kt = (w>>1)&1), half (w>>2) of the block's 128 rows.
// Accumulators: gA persists over the CB column blocks of a region, gSt over its RP row panels
// (same slab scheme as k_grad.hip; gSt gets two slabs per row region, one per row half).
#include <stdlib.h>
#include "pmx_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BG_BM = 128, BG_BN = 64, BG_CB = 4, BG_THREADS = 512;

// ------------------------------------------------------------------------------------------------
// presplit: X (rows x K fp32) -> Xp[3][rowsPad][KP] and Xt[2][KP][rowsPad]  (bf16)
// ------------------------------------------------------------------------------------------------
struct PresplitArgs {
    const float* X[2];
    __bf16* Xp[2];     // [3][rowsPad][KP]
    __bf16* Xt[2];     // [2][KP][rowsPad]
    int64_t rows[2], rowsPad[2];
    int K, KP;
    const DevStatus* status;
};
__global__ __launch_bounds__(256) void k_presplit(PresplitArgs a) {
    __shared__ __bf16 tile[2][64][66];   // [term][row in tile][k]  (+2 pad)
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    const int64_t rows = a.rows[f], rowsPad = a.rowsPad[f];
    const int K = a.K, KP = a.KP;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    if (r0 >= rowsPad) return;
    const float* X = a.X[f];
    __bf16* Xp = a.Xp[f];
    __bf16* Xt = a.Xt[f];
    for (int k0 = 0; k0 < KP; k0 += 64) {
        __syncthreads();
        // 64 rows x 64 k tile: thread handles (row = e / 64, k = e % 64), coalesced over k
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int rr = e >> 6, kk = e & 63;
            const int64_t r = r0 + rr;
            const int k = k0 + kk;
            float x = 0.f;
            if (r < rows && k < K) x = X[r * K + k];
            const __bf16 t0 = (__bf16)x;
            const float e1 = x - (float)t0;
            const __bf16 t1 = (__bf16)e1;
            const __bf16 t2 = (__bf16)(e1 - (float)t1);
            if (k < KP) {
                Xp[((int64_t)0 * rowsPad + r) * KP + k] = t0;
                Xp[((int64_t)1 * rowsPad + r) * KP + k] = t1;
                Xp[((int64_t)2 * rowsPad + r) * KP + k] = t2;
            }
            tile[0][rr][kk] = t0;
            tile[1][rr][kk] = t1;
        }
        __syncthreads();
        // transposed store: thread handles (k = e / 64, row = e % 64), coalesced over rows
        for (int e = threadIdx.x; e < 64 * 64; e += 256) {
            const int kk = e >> 6, rr = e & 63;
            if (k0 + kk < KP) {
                Xt[((int64_t)0 * KP + k0 + kk) * rowsPad + r0 + rr] = tile[0][rr][kk];
                Xt[((int64_t)1 * KP + k0 + kk) * rowsPad + r0 + rr] = tile[1][rr][kk];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct GradBfArgs {
    const float* Y;
    int64_t ldY;
    const __bf16* Ap;    // [3][MPad][KP]
    const __bf16* At;    // [2][KP][MPad]
    const __bf16* Sp;    // [3][NPad][KP]   (terms of St = S^T)
    const __bf16* Stt;   // [2][KP][NPad]   (= terms of S, N contiguous)
    int64_t MPad, NPad;
    float* slabA;        // [nSlabA][M][K]
    float* slabS;        // [nSlabS][N][K]
    double* lossPart;
    const DevStatus* status;
    int M, N, K;
    int RP;
    int doA, doS;        // doA bit 1: ablation switch "no Y traffic" (tuning only)
    int gridX, gridY;
    unsigned long long* prof;   // tuning only: per-phase cycle sums of wave 0 of every workgroup (nullptr = off)
    const float* W;      // M x N weights (ldW) or nullptr for W == 1 (nmf.py:13-41); only the v7 / v8 variants take them
    int64_t ldW;
    const float* absmax; // fp16 two-term path: partial maxima of the factors (k_absmax), nullptr otherwise
    float ymax, wmax;
    int chainL;          // k_grad_f16_v8<.., CHAIN>: see GradV4Args
    unsigned* chainFlags;
    unsigned chainBase;
    DevStatus* wstatus;
    int chainInject;
    float rangeRatio;    // (see GradV4Args)
    int r3;
    int consPrio;        // (see GradV4Args)
};

__device__ __forceinline__ void split2(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

template <int KP, bool EDGE>
__global__ __launch_bounds__(BG_THREADS, 2) void k_grad_bf16(GradBfArgs a) {
    constexpr int KS1 = KP / 16;             // k-steps of GEMM1
    constexpr int LDS_S = KP + 8;            // Sl row stride (bf16 elements): 16-byte multiple, conflict-free b128
    constexpr int LDT_S = BG_BN + 8;         // Stl row stride
    constexpr int LDT_A = BG_BM + 8;         // Atl row stride
    constexpr int LDR = BG_BN + 4;           // Rl row stride (floats)
    // work split of the two gradient contractions over the 8 waves
    constexpr int G2SPLIT = KP == 64 ? 1 : 2;   // KP=32: 4 output tiles x 2 column halves
    constexpr int G3SPLIT = KP == 64 ? 2 : 4;   // row halves / quarters
    constexpr int G2_INNER = BG_BN / G2SPLIT, G3_INNER = BG_BM / G3SPLIT;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* Sl = reinterpret_cast<__bf16*>(smem);                       // [3][64][LDS_S]
    __bf16* Stl = Sl + 3 * BG_BN * LDS_S;                               // [2][KP][LDT_S]
    __bf16* Atl = Stl + 2 * KP * LDT_S;                                 // [2][KP][LDT_A]
    float* Rl = reinterpret_cast<float*>(Atl + 2 * KP * LDT_A);         // [128][LDR]
    float* Yl = Rl + BG_BM * LDR;                                       // [8 waves][32][32]  private Y landing tiles

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int M = a.M, N = a.N, K = a.K;
    // Workgroup -> region map.  Consecutive workgroup ids land on consecutive XCDs (id % 8), each with its own
    // 4 MiB L2.  The S terms a region streams (46 KB per step) are shared by all row regions of the same column
    // region, so every XCD is given a contiguous eighth of the column regions: its share of the S terms then
    // stays L2-resident instead of being re-fetched across the fabric.  (Speed only; any placement is correct.)
    int rowRegion, colRegion;
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if (gy % 8 == 0) {
            const int xcd = lin & 7, idx = lin >> 3;
            rowRegion = idx % gx;
            colRegion = xcd * (gy >> 3) + idx / gx;
        } else {
            rowRegion = lin % gx;
            colRegion = lin / gx;
        }
    }
    const int row0 = rowRegion * a.RP * BG_BM;
    const int col0 = colRegion * BG_CB * BG_BN;

    const int g1_mt = w >> 1, g1_nt = w & 1;
    const int g2_mt = w >> 1;
    const int g2_kt = KP == 64 ? (w & 1) : 0;
    const int g2_half = KP == 64 ? 0 : (w & 1);
    const int g3_nt = w & 1;
    const int g3_kt = KP == 64 ? ((w >> 1) & 1) : 0;
    const int g3_part = KP == 64 ? (w >> 2) : (w >> 1);

    f32x16 accS[BG_CB];
#pragma unroll
    for (int cb = 0; cb < BG_CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accS[cb][i] = 0.f;
    f32x16 accA;
    f32x16 p;
    bf16x8 afr[KS1][3];
    float lossAcc = 0.f;

    int nrp = (M - row0 + BG_BM - 1) / BG_BM;
    if (nrp > a.RP) nrp = a.RP;
    int ncb = (N - col0 + BG_BN - 1) / BG_BN;
    if (ncb > BG_CB) ncb = BG_CB;
    const int nsteps = nrp * ncb;

    const int laneRow = g1_mt * 32 + 4 * hi;
    const int laneCol = g1_nt * 32 + l31;
    const int laneOff = laneRow * (int)a.ldY + laneCol;
    // Y tile of this wave (32 x 32 fp32).  !EDGE: fetched by LDS-DMA (global_load_lds, 16 B per lane, 8 rows per
    // instruction) into a wave-private 4 KiB landing tile -- no registers are tied up while it is in flight, so
    // the request for step s+1 is issued before GEMM1 of step s and has a whole step to arrive.  The tile is
    // produced and consumed by the same wave: its own vmcnt wait is the only synchronisation needed.
    // EDGE (partial blocks): guarded register loads issued after the residual has been parked.
    float* Ytile = Yl + w * 1024;
    const int dmaRow = lane >> 3, dmaCol = (lane & 7) * 4;     // lane -> (row within 8, first of 4 columns)
    // The DMA is issued from inline asm so that the compiler does not know LDS is being written behind its back:
    // with the builtin it conservatively drains vmcnt before the next ds_read and before every barrier, which
    // serialises the transfer.  Completion is guaranteed by program order instead (see the wait at its consumer).
    const unsigned ytile_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)Ytile;
    auto request_Y_dma = [&](int prow0, int bcol0) {
        const float* src = a.Y + (int64_t)(prow0 + g1_mt * 32 + dmaRow) * a.ldY + bcol0 + g1_nt * 32 + dmaCol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* g = src + (int64_t)q * 8 * a.ldY;
            const unsigned dst = __builtin_amdgcn_readfirstlane(ytile_lds + q * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    };
    auto request_Y = [&](int prow0, int bcol0) {
        const float* blk = a.Y + (int64_t)prow0 * a.ldY + bcol0;
        if (prow0 + BG_BM <= M && bcol0 + BG_BN <= N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* rowp = blk + (int64_t)((i & 3) + 8 * (i >> 2)) * a.ldY;
                p[i] = rowp[laneOff];
            }
        } else {
            const int rmax = M - 1 - prow0, cmax = N - 1 - bcol0;
            const int cc = laneCol < cmax ? laneCol : cmax;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lr = laneRow + (i & 3) + 8 * (i >> 2);
                const int rr = lr < rmax ? lr : rmax;
                const float v = blk[(int64_t)rr * a.ldY + cc];
                p[i] = (lr <= rmax && laneCol <= cmax) ? v : 0.f;
            }
        }
    };
    auto flush_gA = [&](int prow0) {
        const int slab = colRegion * G2SPLIT + g2_half;
        float* dst = a.slabA + (int64_t)slab * M * K;
        const int kk = g2_kt * 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gr = prow0 + g2_mt * 32 + tile_row(i, lane);
            if (gr < M && kk < K) dst[(int64_t)gr * K + kk] = accA[i];
        }
    };
    // S terms: 16-byte unguarded copies from the zero-padded presplit arrays.  They are requested into registers
    // one step ahead (right after this step's operands became visible) and written to LDS at the top of the next
    // step, so the L2 round trip overlaps the three contractions.
    constexpr int CH = KP / 8;                                   // 16-byte chunks per Sl row
    constexpr int T_SL = 3 * BG_BN * CH, T_STL = 2 * KP * 8;      // chunk counts
    constexpr int N_SL = (T_SL + BG_THREADS - 1) / BG_THREADS, N_STL = (T_STL + BG_THREADS - 1) / BG_THREADS;
    uint4 pre_sl[N_SL], pre_stl[N_STL];
#define BG_LOAD_S(BCOL0)                                                                                         \
    do {                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < N_SL; ++u) {                                                       \
            const int e = tid + u * BG_THREADS;                                                                  \
            const int t = e / (BG_BN * CH), r = (e / CH) % BG_BN, c = e % CH;                                    \
            if (T_SL % BG_THREADS == 0 || e < T_SL)                                                              \
                pre_sl[u] = *reinterpret_cast<const uint4*>(a.Sp + ((int64_t)t * a.NPad + (BCOL0) + r) * KP + c * 8); \
        }                                                                                                        \
        _Pragma("unroll") for (int u = 0; u < N_STL; ++u) {                                                      \
            const int e = tid + u * BG_THREADS;                                                                  \
            const int t = e / (KP * 8), r = (e / 8) % KP, c = e % 8;                                             \
            if (T_STL % BG_THREADS == 0 || e < T_STL)                                                            \
                pre_stl[u] = *reinterpret_cast<const uint4*>(a.Stt + ((int64_t)t * KP + r) * a.NPad + (BCOL0) + c * 8); \
        }                                                                                                        \
    } while (0)
#define BG_STORE_S()                                                                                             \
    do {                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < N_SL; ++u) {                                                       \
            const int e = tid + u * BG_THREADS;                                                                  \
            const int t = e / (BG_BN * CH), r = (e / CH) % BG_BN, c = e % CH;                                    \
            if (T_SL % BG_THREADS == 0 || e < T_SL)                                                              \
                *reinterpret_cast<uint4*>(Sl + (t * BG_BN + r) * LDS_S + c * 8) = pre_sl[u];                     \
        }                                                                                                        \
        _Pragma("unroll") for (int u = 0; u < N_STL; ++u) {                                                      \
            const int e = tid + u * BG_THREADS;                                                                  \
            const int t = e / (KP * 8), r = (e / 8) % KP, c = e % 8;                                             \
            if (T_STL % BG_THREADS == 0 || e < T_STL)                                                            \
                *reinterpret_cast<uint4*>(Stl + (t * KP + r) * LDT_S + c * 8) = pre_stl[u];                      \
        }                                                                                                        \
    } while (0)
    auto stage_A = [&](int prow0) {
        // Atl[t][kk][m]: rows of 128 bf16 (16 chunks)
        for (int e = tid; e < 2 * KP * 16; e += BG_THREADS) {
            const int t = e / (KP * 16), r = (e / 16) % KP, c = e % 16;
            const uint4 v = *reinterpret_cast<const uint4*>(a.At + ((int64_t)t * KP + r) * a.MPad + prow0 + c * 8);
            *reinterpret_cast<uint4*>(Atl + (t * KP + r) * LDT_A + c * 8) = v;
        }
        // this wave's GEMM1 A fragments, straight from global into registers
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                afr[ks][t] = *reinterpret_cast<const bf16x8*>(a.Ap + ((int64_t)t * a.MPad + prow0 + g1_mt * 32 + l31) * KP + ks * 16 + hi * 8);
    };

    const bool noY = (a.doA & 2) != 0;
    if (nsteps > 0) {
        if (!noY) {
            if (EDGE) request_Y(row0, col0);
            else request_Y_dma(row0, col0);
        }
    }
    int rp = 0, cb = 0;
#pragma nounroll
    for (int step = 0; step < nsteps; ++step) {
        const int prow0 = row0 + rp * BG_BM;
        int nrp_ = rp, ncb_ = cb + 1;
        if (ncb_ == ncb) { ncb_ = 0; nrp_ = rp + 1; }
        const bool more = step + 1 < nsteps;
        __syncthreads();                       // B0: previous step's LDS readers are done
        if (cb == 0) {
            stage_A(prow0);
#pragma unroll
            for (int i = 0; i < 16; ++i) accA[i] = 0.f;
        }
        BG_LOAD_S(col0 + cb * BG_BN);          // (holding these 20 registers across a step costs scratch spills:
        BG_STORE_S();                          //  the S terms are staged synchronously, the Y tile asynchronously)
        __syncthreads();                       // B1: staged operands visible
        // ---- P accumulator starts at -Y -----------------------------------------------------------------
        if (!EDGE) {
            if (noY) {
#pragma unroll
                for (int i = 0; i < 16; ++i) p[i] = 0.f;
            } else {
                // the landing tile was requested a step ago; the S-term loads consumed above were issued after it
                // and memory operations retire in order, so it has landed (the explicit wait costs nothing)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) p[i] = -Ytile[tile_row(i, lane) * 32 + l31];
                if (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tile has been read: it may be overwritten
                    request_Y_dma(row0 + nrp_ * BG_BM, col0 + ncb_ * BG_BN);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = -p[i];
        }
        // ---- GEMM1: P = A S on top of -Y, 6 products per k-step ------------------------------------
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const __bf16* sb = Sl + (g1_nt * 32 + l31) * LDS_S + ks * 16 + hi * 8;
            const bf16x8 s0 = *reinterpret_cast<const bf16x8*>(sb);
            const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(sb + BG_BN * LDS_S);
            const bf16x8 s2 = *reinterpret_cast<const bf16x8*>(sb + 2 * BG_BN * LDS_S);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][2], s0, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], s1, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], s2, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], s0, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], s1, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], s0, p, 0, 0, 0);
        }
        // ---- loss, park R (fp32) in LDS ---------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lr = g1_mt * 32 + tile_row(i, lane);
            const float r = p[i];
            lossAcc += r * r;
            Rl[lr * LDR + g1_nt * 32 + l31] = r;
        }
        // B2: R visible.  Raw barrier: the LDS-DMA of the next Y tile is in flight and must stay in flight
        // (a __syncthreads() here would make the compiler drain vmcnt first)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (EDGE && more && !noY) request_Y(row0 + nrp_ * BG_BM, col0 + ncb_ * BG_BN);
        // ---- GEMM2: gA tile += R . St_blk   (contraction over this wave's share of the 64 columns)
        //      GEMM3: gSt tile += R^T . A_panel (contraction over this wave's share of the 128 rows)
        // The two have independent accumulators; their k-steps are interleaved in one basic block so that each
        // MFMA chain's dependency latency and the other's LDS reads / bf16 splits overlap.
#define BG_G2_STEP(KS)                                                                                  \
    {                                                                                                   \
        const int n0 = g2_half * G2_INNER + (KS) * 16 + hi * 8;                                         \
        const float* rrow = Rl + (g2_mt * 32 + l31) * LDR + n0;                                         \
        float x[8];                                                                                     \
        *reinterpret_cast<float4*>(&x[0]) = *reinterpret_cast<const float4*>(rrow);                     \
        *reinterpret_cast<float4*>(&x[4]) = *reinterpret_cast<const float4*>(rrow + 4);                 \
        bf16x8 r0, r1;                                                                                  \
        split2(x, r0, r1);                                                                              \
        const __bf16* sb = Stl + (g2_kt * 32 + l31) * LDT_S + n0;                                       \
        const bf16x8 s0 = *reinterpret_cast<const bf16x8*>(sb);                                         \
        const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(sb + KP * LDT_S);                            \
        accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, s0, accA, 0, 0, 0);                          \
        accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s1, accA, 0, 0, 0);                          \
        accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s0, accA, 0, 0, 0);                          \
    }
#define BG_G3_STEP(ACC, KS)                                                                             \
    {                                                                                                   \
        const int m0 = g3_part * G3_INNER + (KS) * 16 + hi * 8;                                         \
        const float* rcol = Rl + m0 * LDR + g3_nt * 32 + l31;                                           \
        float x[8];                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) x[q] = rcol[q * LDR];                              \
        bf16x8 r0, r1;                                                                                  \
        split2(x, r0, r1);                                                                              \
        const __bf16* ab = Atl + (g3_kt * 32 + l31) * LDT_A + m0;                                       \
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ab);                                         \
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ab + KP * LDT_A);                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, a0, ACC, 0, 0, 0);                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, a1, ACC, 0, 0, 0);                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, a0, ACC, 0, 0, 0);                            \
    }
        static_assert(G2_INNER / 16 == G3_INNER / 16 || true, "");
        constexpr int NK2 = G2_INNER / 16, NK3 = G3_INNER / 16, NKM = NK2 > NK3 ? NK2 : NK3;
        const bool wantA = (a.doA & 1) != 0, wantS = a.doS != 0;
#define BG_BOTH_INTO(ACC)                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < NKM; ++ks) {                                                \
        if (wantA && ks < NK2) BG_G2_STEP(ks)                                                           \
        if (wantS && ks < NK3) BG_G3_STEP(ACC, ks)                                                      \
    }
        switch (cb) {
            case 0: BG_BOTH_INTO(accS[0]) break;
            case 1: BG_BOTH_INTO(accS[1]) break;
            case 2: BG_BOTH_INTO(accS[2]) break;
            default: BG_BOTH_INTO(accS[3]) break;
        }
#undef BG_BOTH_INTO
#undef BG_G2_STEP
#undef BG_G3_STEP
        if (cb + 1 == ncb && (a.doA & 1)) flush_gA(prow0);
        cb = ncb_;
        rp = nrp_;
    }
    // ---- flush gSt: slab = rowRegion * G3SPLIT + part ---------------------------------------------------
    if (a.doS) {
        const int slab = rowRegion * G3SPLIT + g3_part;
        float* dst = a.slabS + (int64_t)slab * N * K;
        const int kk = g3_kt * 32 + l31;
#pragma unroll
        for (int cbi = 0; cbi < BG_CB; ++cbi) {
            const int bcol0 = col0 + cbi * BG_BN;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gn = bcol0 + g3_nt * 32 + tile_row(i, lane);
                if (gn < N && kk < K) dst[(int64_t)gn * K + kk] = accS[cbi][i];
            }
        }
    }
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[w] = v;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < BG_THREADS / 64; ++i) s += (double)red[i];
            a.lossPart[blockIdx.x] = s;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// k_grad_bf16_pipe: the same computation for whole-block shapes (M % 128 == 0, N % 64 == 0, 16-byte aligned Y
// rows), software-pipelined with LDS-DMA so that no global-memory round trip is exposed inside a step:
//   * the Y tile of step s+1 lands in a wave-private tile, requested right after the tile of step s was read;
//   * the S terms of step s+1 are DMA'd into the SAME single LDS images as soon as all waves are done reading
//     them: Sl (read only by GEMM1) right after barrier B2, Stl (read only by GEMM2) right after barrier B3.
//     Padded row strides are kept (bank-conflict-free ds_read_b128): the images are transferred as linear
//     1 KiB chunks, one per wave instruction, and the lanes that fall on a pad slot fetch a dummy address.
//   * all DMA is issued from inline asm (invisible to the compiler's wait-count model, which would otherwise
//     drain it before every LDS read and barrier); completion is enforced with hand-counted s_waitcnt vmcnt(N)
//     in front of the raw s_barrier that publishes the data.  Per wave and step, in issue order:
//         Y(s+1): 4 instructions   Sl(s+1): NI_SL   Stl(s+1): NI_STL
//     top of step s+1:  vmcnt(NI_STL)  -> Y, Sl landed, Stl may still fly;   B2:  vmcnt(4) -> Stl landed.
// Registers hold no in-flight data, so nothing spills.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

template <int KP, bool PROF>
__global__ __launch_bounds__(BG_THREADS, 2) void k_grad_bf16_pipe(GradBfArgs a) {
    constexpr int KS1 = KP / 16;
    constexpr int LDS_S = KP + 8, LDT_S = BG_BN + 8, LDT_A = BG_BM + 8, LDR = BG_BN + 4;
    constexpr int G2SPLIT = KP == 64 ? 1 : 2, G3SPLIT = KP == 64 ? 2 : 4;
    constexpr int G2_INNER = BG_BN / G2SPLIT, G3_INNER = BG_BM / G3SPLIT;
    constexpr int NW = BG_THREADS / 64;
    // linear DMA images
    constexpr int SL_ROWB = LDS_S * 2, SL_TERMB = BG_BN * SL_ROWB, SL_BYTES = 3 * SL_TERMB, SL_DATA = KP / 8;
    constexpr int STL_ROWB = LDT_S * 2, STL_TERMB = KP * STL_ROWB, STL_BYTES = 2 * STL_TERMB, STL_DATA = BG_BN / 8;
    constexpr int NI_SL = (SL_BYTES + 1024 * NW - 1) / (1024 * NW);      // DMA instructions per wave
    constexpr int NI_STL = (STL_BYTES + 1024 * NW - 1) / (1024 * NW);
    constexpr int SL_REGION = NI_SL * NW * 1024, STL_REGION = NI_STL * NW * 1024;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* Sl = reinterpret_cast<__bf16*>(smem);
    __bf16* Stl = reinterpret_cast<__bf16*>(smem + SL_REGION);
    __bf16* Atl = reinterpret_cast<__bf16*>(smem + SL_REGION + STL_REGION);      // [2][KP][LDT_A]
    float* Rl = reinterpret_cast<float*>(Atl + 2 * KP * LDT_A);                  // [128][LDR]
    float* Yl = Rl + BG_BM * LDR;                                                // [8 waves][32][32]

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int M = a.M, N = a.N, K = a.K;
    int rowRegion, colRegion;
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if (gy % 8 == 0) {
            const int xcd = lin & 7, idx = lin >> 3;
            rowRegion = idx % gx;
            colRegion = xcd * (gy >> 3) + idx / gx;
        } else {
            rowRegion = lin % gx;
            colRegion = lin / gx;
        }
    }
    const int row0 = rowRegion * a.RP * BG_BM;
    const int col0 = colRegion * BG_CB * BG_BN;

    const int g1_mt = w >> 1, g1_nt = w & 1;
    const int g2_mt = w >> 1;
    const int g2_kt = KP == 64 ? (w & 1) : 0;
    const int g2_half = KP == 64 ? 0 : (w & 1);
    const int g3_nt = w & 1;
    const int g3_kt = KP == 64 ? ((w >> 1) & 1) : 0;
    const int g3_part = KP == 64 ? (w >> 2) : (w >> 1);

    f32x16 accS[BG_CB];
#pragma unroll
    for (int cb = 0; cb < BG_CB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accS[cb][i] = 0.f;
    f32x16 accA;
    f32x16 p;
    bf16x8 afr[KS1][3];
    float lossAcc = 0.f;

    int nrp = (M - row0 + BG_BM - 1) / BG_BM;
    if (nrp > a.RP) nrp = a.RP;
    int ncb = (N - col0 + BG_BN - 1) / BG_BN;
    if (ncb > BG_CB) ncb = BG_CB;
    const int nsteps = nrp * ncb;
    const bool noY = (a.doA & 2) != 0;

    // ---- DMA address tables (step-invariant per-lane element offsets) ------------------------------------
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    int goff_sl[NI_SL], goff_stl[NI_STL];
#pragma unroll
    for (int i = 0; i < NI_SL; ++i) {
        const int o = 1024 * (i * NW + w) + 16 * lane;
        const int t = o / SL_TERMB, wi = o % SL_TERMB, r = wi / SL_ROWB, sl = (wi % SL_ROWB) / 16;
        goff_sl[i] = (o < SL_BYTES && sl < SL_DATA) ? (int)(((int64_t)t * a.NPad + r) * KP + sl * 8) : 0;
    }
#pragma unroll
    for (int i = 0; i < NI_STL; ++i) {
        const int o = 1024 * (i * NW + w) + 16 * lane;
        const int t = o / STL_TERMB, wi = o % STL_TERMB, r = wi / STL_ROWB, sl = (wi % STL_ROWB) / 16;
        goff_stl[i] = (o < STL_BYTES && sl < STL_DATA) ? (int)(((int64_t)t * KP + r) * a.NPad + sl * 8) : 0;
    }
    auto dma_Sl = [&](int bcol0) {
        const __bf16* base = a.Sp + (int64_t)bcol0 * KP;
#pragma unroll
        for (int i = 0; i < NI_SL; ++i)
            lds_dma16(base + goff_sl[i], __builtin_amdgcn_readfirstlane(lds_base + 1024 * (i * NW + w)));
    };
    auto dma_Stl = [&](int bcol0) {
        const __bf16* base = a.Stt + bcol0;
#pragma unroll
        for (int i = 0; i < NI_STL; ++i)
            lds_dma16(base + goff_stl[i], __builtin_amdgcn_readfirstlane(lds_base + SL_REGION + 1024 * (i * NW + w)));
    };
    float* Ytile = Yl + w * 1024;
    const unsigned ytile_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)Ytile;
    const int dmaRow = lane >> 3, dmaCol = (lane & 7) * 4;
    auto dma_Y = [&](int prow0, int bcol0) {
        const float* src = a.Y + (int64_t)(prow0 + g1_mt * 32 + dmaRow) * a.ldY + bcol0 + g1_nt * 32 + dmaCol;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            lds_dma16(src + (int64_t)q * 8 * a.ldY, __builtin_amdgcn_readfirstlane(ytile_lds + q * 1024));
    };
    auto flush_gA = [&](int prow0) {
        const int slab = colRegion * G2SPLIT + g2_half;
        float* dst = a.slabA + (int64_t)slab * M * K;
        const int kk = g2_kt * 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gr = prow0 + g2_mt * 32 + tile_row(i, lane);
            if (kk < K) dst[(int64_t)gr * K + kk] = accA[i];
        }
    };
    auto stage_A = [&](int prow0) {
        constexpr int CA = BG_BM / 8;
        for (int e = tid; e < 2 * KP * CA; e += BG_THREADS) {
            const int t = e / (KP * CA), r = (e / CA) % KP, c = e % CA;
            const uint4 v = *reinterpret_cast<const uint4*>(a.At + ((int64_t)t * KP + r) * a.MPad + prow0 + c * 8);
            *reinterpret_cast<uint4*>(Atl + (t * KP + r) * LDT_A + c * 8) = v;
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                afr[ks][t] = *reinterpret_cast<const bf16x8*>(a.Ap + ((int64_t)t * a.MPad + prow0 + g1_mt * 32 + l31) * KP + ks * 16 + hi * 8);
        // Consume the loads HERE: otherwise the compiler's wait for them lands in front of GEMM1 on every step
        // (it cannot see that the panel switch is rare) and, blind to the asm DMAs, drains the Y prefetch with it.
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(afr[ks][t]));
    };

    if (nsteps > 0) {   // prologue: same issue order as inside the loop
        if (!noY) dma_Y(row0, col0);
        else { lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds));
               lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); }
        dma_Sl(col0);
        dma_Stl(col0);
    }
    // phase profiler (PROF instantiation only: its counters cost ~20 VGPRs, which the production kernel needs)
    unsigned long long ph[PROF ? 10 : 1] = {};
    const bool prof = PROF && a.prof != nullptr && w == 0;
#define PH(i) if constexpr (PROF) { if (prof) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tprev; tprev = t_; } }
    unsigned long long tprev = prof ? __builtin_readcyclecounter() : 0;
    int rp = 0, cb = 0;
#pragma nounroll
    for (int step = 0; step < nsteps; ++step) {
        const int prow0 = row0 + rp * BG_BM;
        int nrp_ = rp, ncb_ = cb + 1;
        if (ncb_ == ncb) { ncb_ = 0; nrp_ = rp + 1; }
        const bool more = step + 1 < nsteps;
        const int nprow0 = row0 + nrp_ * BG_BM, nbcol0 = col0 + ncb_ * BG_BN;
        // ---- B0: previous step done everywhere; Y(s) and Sl(s) landed (Stl(s) may still be in flight) ---------
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NI_STL) : "memory");
        __builtin_amdgcn_s_barrier();
        PH(0)
        if (cb == 0) {
            stage_A(prow0);
#pragma unroll
            for (int i = 0; i < 16; ++i) accA[i] = 0.f;
        }
        PH(1)
        // ---- P accumulator starts at -Y; then request the next Y tile into the same private tile ---------------
        if (noY) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = -Ytile[tile_row(i, lane) * 32 + l31];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more && !noY) dma_Y(nprow0, nbcol0);
        else {   // keep the per-step instruction count fixed so the hand-counted waits stay valid
            lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds));
            lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds));
        }
        PH(2)
        // ---- GEMM1 ------------------------------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const __bf16* sb = Sl + (g1_nt * 32 + l31) * LDS_S + ks * 16 + hi * 8;
            const bf16x8 s0 = *reinterpret_cast<const bf16x8*>(sb);
            const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(sb + BG_BN * LDS_S);
            const bf16x8 s2 = *reinterpret_cast<const bf16x8*>(sb + 2 * BG_BN * LDS_S);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][2], s0, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], s1, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], s2, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], s0, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], s1, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], s0, p, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lr = g1_mt * 32 + tile_row(i, lane);
            const float r = p[i];
            lossAcc += r * r;
            Rl[lr * LDR + g1_nt * 32 + l31] = r;
        }
        PH(3)
        // ---- B2: R visible, Stl(s) landed (the 4 Y instructions issued above may still fly) -----------------
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PH(4)
        dma_Sl(more ? nbcol0 : col0);          // every wave is done with GEMM1's reads of Sl
        // ---- GEMM2 ------------------------------------------------------------------------------------------
        if (a.doA & 1) {
#pragma unroll
            for (int ks = 0; ks < G2_INNER / 16; ++ks) {
                const int n0 = g2_half * G2_INNER + ks * 16 + hi * 8;
                const float* rrow = Rl + (g2_mt * 32 + l31) * LDR + n0;
                float x[8];
                *reinterpret_cast<float4*>(&x[0]) = *reinterpret_cast<const float4*>(rrow);
                *reinterpret_cast<float4*>(&x[4]) = *reinterpret_cast<const float4*>(rrow + 4);
                bf16x8 r0, r1;
                split2(x, r0, r1);
                const __bf16* sb = Stl + (g2_kt * 32 + l31) * LDT_S + n0;
                const bf16x8 s0 = *reinterpret_cast<const bf16x8*>(sb);
                const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(sb + KP * LDT_S);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, s0, accA, 0, 0, 0);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s1, accA, 0, 0, 0);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s0, accA, 0, 0, 0);
            }
        }
        PH(5)
        // ---- B3: every wave is done with GEMM2's reads of Stl ---------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PH(6)
        dma_Stl(more ? nbcol0 : col0);
        // ---- GEMM3 ------------------------------------------------------------------------------------------
        if (a.doS) {
#define BG_GEMM3_INTO(ACC)                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < G3_INNER / 16; ++ks) {                                     \
        const int m0 = g3_part * G3_INNER + ks * 16 + hi * 8;                                           \
        const float* rcol = Rl + m0 * LDR + g3_nt * 32 + l31;                                           \
        float x[8];                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) x[q] = rcol[q * LDR];                              \
        bf16x8 r0, r1;                                                                                  \
        split2(x, r0, r1);                                                                              \
        const __bf16* ab = Atl + (g3_kt * 32 + l31) * LDT_A + m0;                                       \
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ab);                                         \
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ab + KP * LDT_A);                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, a0, ACC, 0, 0, 0);                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, a1, ACC, 0, 0, 0);                            \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, a0, ACC, 0, 0, 0);                            \
    }
            switch (cb) {
                case 0: BG_GEMM3_INTO(accS[0]) break;
                case 1: BG_GEMM3_INTO(accS[1]) break;
                case 2: BG_GEMM3_INTO(accS[2]) break;
                default: BG_GEMM3_INTO(accS[3]) break;
            }
#undef BG_GEMM3_INTO
        }
        PH(7)
        if (cb + 1 == ncb && (a.doA & 1)) flush_gA(prow0);
        PH(8)
        cb = ncb_;
        rp = nrp_;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the DMAs of the (non-existent) step after the last
    if (a.doS) {
        const int slab = rowRegion * G3SPLIT + g3_part;
        float* dst = a.slabS + (int64_t)slab * N * K;
        const int kk = g3_kt * 32 + l31;
#pragma unroll
        for (int cbi = 0; cbi < BG_CB; ++cbi) {
            const int bcol0 = col0 + cbi * BG_BN;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int gn = bcol0 + g3_nt * 32 + tile_row(i, lane);
                if (gn < N && kk < K) dst[(int64_t)gn * K + kk] = accS[cbi][i];
            }
        }
    }
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[w] = v;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < BG_THREADS / 64; ++i) s += (double)red[i];
            a.lossPart[blockIdx.x] = s;
        }
    }
    PH(9)
    if constexpr (PROF) {
        if (prof && lane == 0)
            for (int i = 0; i < 10; ++i) atomicAdd(&a.prof[i], ph[i]);
    }
#undef PH
}


// ------------------------------------------------------------------------------------------------
// Shared by the fp32-operand variants below: the XOR chunk swizzle of the 128-byte-row LDS images and the transposing
// read.  (They carry the name of the 64 x 64 / two-workgroups-per-CU variant `v3` they were written for; that variant
// and `v6` -- v7 with LDS-DMA landing tiles for Y and per-panel S staging -- were measured, superseded and removed:
// DESIGN.md section 4 keeps their numbers.)
// ------------------------------------------------------------------------------------------------
// chunk permutation of the R images: 16-byte chunk c of row n lives at chunk c ^ v3_swz(n)
__device__ __forceinline__ int v3_swz(int n) {
    const int x = (n >> 1) & 7;
    return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1);
}
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 v3_tr_pair(const unsigned char* base, int off0, int off1) {
    // two transposing reads = the 8 contraction slots of one 32x32x16 operand
    const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off0));
    const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off1));
    bf16x8 r;
    r[0] = a0[0]; r[1] = a0[1]; r[2] = a0[2]; r[3] = a0[3];
    r[4] = a1[0]; r[5] = a1[1]; r[6] = a1[2]; r[7] = a1[3];
    return r;
}


// ------------------------------------------------------------------------------------------------
// k_grad_bf16_v4 (K = 64, whole 128 x 64 blocks): the minimum-traffic variant.
//
// In the LDS-DMA variants above most of the bytes a CU pulls per step are OPERANDS, not Y: the bf16 terms of S
// (6-10 bytes per fp32 element, re-fetched for every row panel) and of A, and the waves stall at the DMA issue.
// This variant fetches every operand as the fp32 it is (4 bytes per element) and splits it into bf16 terms
// inside the kernel, so per 128 x 64 step a CU pulls 32 KiB of Y + 16 KiB of S (+ 32 KiB of A per 4 steps)
// instead of ~100 KiB; no presplit pass, no second orientation of anything:
//   * S block (64 x 64 fp32, one contiguous 16 KiB run of St) and A panel (128 x 64 fp32, 32 KiB) are loaded
//     one step ahead into registers (8 / 16 VGPRs per thread), split into 3 bf16 terms and written to the
//     [row][k] images Sl (double-buffered) and Aimg; the transposed operands GEMM2/GEMM3 need are produced
//     by the transposing LDS read (ds_read_b64_tr_b16) from those same images;
//   * Y is loaded straight into the accumulator layout, one step ahead (16 VGPRs), no LDS round trip;
//   * R is parked once as two bf16 [n][m] images (see v3) and never converted again.
// All global loads are ordinary loads, so the compiler's wait counts are exact; no inline-asm DMA here.
// Two barriers per step: TOP (publishes Sl(s), retires R(s-1)) and B_R (publishes R(s)).
// ------------------------------------------------------------------------------------------------
constexpr int V4_BM = 128, V4_BN = 64, V4_THREADS = 512, V4_NW = 8;
constexpr int V4_SL_BYTES = 3 * 64 * 144, V4_A_BYTES = 3 * 128 * 144, V4_R_TERM = 64 * 256;
constexpr int V4_OFF_A = 2 * V4_SL_BYTES, V4_OFF_R = V4_OFF_A + V4_A_BYTES, V4_LDS_BYTES = V4_OFF_R + 2 * V4_R_TERM;
static_assert(V4_OFF_R % 256 == 0, "R images must start on a bank row");
static_assert(V4_LDS_BYTES <= 160 * 1024, "");

// chunk permutation of the 256-byte-row R images: 16-byte chunk c of row n lives at chunk c ^ v4_swz(n)
__device__ __forceinline__ int v4_swz(int n) { return ((n & 3) << 2) | ((n >> 2) & 3); }

__device__ __forceinline__ void v4_split3(const float4& x, bf16x4& t0, bf16x4& t1, bf16x4& t2) {
    const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float e1 = v[j] - (float)a;
        const __bf16 b = (__bf16)e1;
        t0[j] = a;
        t1[j] = b;
        t2[j] = (__bf16)(e1 - (float)b);
    }
}

struct GradV4Args {
    const float* Y;
    int64_t ldY;
    const float* A;      // [M][64]
    const float* St;     // [N][64]
    float* slabA;
    float* slabS;
    double* lossPart;
    const DevStatus* status;
    int M, N;
    int RP;
    int doA, doS;
    int gridX, gridY;
    unsigned long long* prof;
    const float* W;      // M x N weights (ldW) or nullptr; only k_grad_bf16_v7 / k_grad_f16_v8 take them
    int64_t ldW;
    const float* absmax; // k_grad_f16_v8: [2][V8_NPART] partial maxima of |A|, |St| (k_absmax)
    float ymax;          // k_grad_f16_v8: max |Y|
    float wmax;          // k_grad_f16_v8: max(1, max |W|) (1 without weights): D = W R must fit fp16 as well
    // k_grad_f16_v8<.., CHAIN>: gA accumulated in place through the XCD's L2 by chains of chainL workgroups (see the kernel)
    int chainL;          // workgroups per chain (0: slabs, one per column region)
    unsigned* chainFlags;   // [chains][RP][4] arrival words, monotonic over launches
    unsigned chainBase;  // launch sequence number * 64
    DevStatus* wstatus;  // writable view of `status` (fault report)
    int chainInject;     // tests: report a fault from this launch (exercises the host's fall-back)
    float rangeRatio;    // [r4] two-term fp16 kernels: report k1_fault 4 when K max|A| max|S| > rangeRatio max|Y| (0: no check; f16_range_fault)
    int r3;              // [r4] k_grad_f16_v8<.., R3>: third terms of A and S in the residual's product, two accumulators
    int consPrio;        // [r6] s_setprio level of the CONSUMER waves for the launch (0: none): the consumers are the pole of every slot, the producers wait ~30 % of it at the barrier
    K1GramFold fold;     // [r6] k_grad_f16_k32: the step rule's Gram fold riding in the first workgroups (pmx_common.h)
};

// (k_grad_bf16_v4's kernel was removed in round 4 together with k_grad_bf16_v5's: with K1's zero-padded frame -- pmx_k1_frame -- the
// shapes they served, M % 128 = 0 with N % 64 = 0 but not N % 256 = 0, run k_grad_bf16_v7 / k_grad_f16_v8 on columns rounded up, or
// the guarded kernel where that would cost more than a quarter more entries; the constants and helpers above are v7's as well)


// ------------------------------------------------------------------------------------------------
// k_grad_bf16_v5 (K = 64, M % 128 == 0, N % 32 == 0): producer / consumer wavefronts.
//
// In every variant above all waves of a workgroup walk through the same phases together (operand reads, a
// dependent MFMA chain, a barrier), so the matrix pipe idles while the LDS works and vice versa.  Here the 8 waves
// of a workgroup split into two roles that are busy with DIFFERENT things at the same time, one of each per SIMD:
//   producers (waves 0-3)  slot t: P = A S - Y for the 128 x 32 block t (wave j: rows 32j..32j+31; its A terms
//                          live in registers for the whole row panel), split R into two bf16 terms, park it in
//                          R[t & 1]
//   consumers (waves 4-7)  slot t: the two gradient contractions of block t-1 from R[(t-1) & 1]:
//                          gA (wave c: rows 32c.., both 32-wide k tiles) and gSt (wave c: k tile c & 1, row half c >> 1)
// One barrier per slot.  24 + 24 MFMAs per SIMD-pair of waves per slot, operands: S terms (Sl, triple-buffered:
// written in slot t-1 by all waves from fp32 St, read by the producers in slot t and by the consumers, through the
// transposing read, in slot t+1), A terms for gSt (Aimg, double-buffered per row panel, written by the producers
// from their register fragments), R (see v4: swizzled [n][m] bf16 images).  The consumers carry no A/S fragments
// between slots, so they hold 8 column blocks of gSt accumulators: regions are 256 columns wide with 32-column
// steps.  Everything is plain loads (exact compiler wait counts); Y and St/A are requested one slot ahead.
// ------------------------------------------------------------------------------------------------
constexpr int V5_BM = 128, V5_BN = 32, V5_NB = 8, V5_THREADS = 512;
constexpr int V5_S_TERM = 32 * 128, V5_SL_BYTES = 3 * V5_S_TERM, V5_A_TERM = 128 * 128, V5_AIMG_BYTES = 2 * V5_A_TERM,
              V5_R_TERM = 32 * 256, V5_R_BYTES = 2 * V5_R_TERM;
constexpr int V5_OFF_A = 3 * V5_SL_BYTES, V5_OFF_R = V5_OFF_A + V5_AIMG_BYTES, V5_OFF_Y = V5_OFF_R + 2 * V5_R_BYTES,
              V5_LDS_BYTES = V5_OFF_Y + 4 * 4096;
static_assert(V5_OFF_R % 256 == 0, "R images must start on a bank row");
static_assert(V5_LDS_BYTES <= 160 * 1024, "");
static_assert(V5_NB * V5_BN == BG_CB * BG_BN, "same region width as the other variants (shared plan)");

// (kernel removed in round 4, see above: what is left of v5 is its frame -- the constants, roles and image layouts k_grad_bf16_v7,
// k_grad_f16_v8, k_grad_f16_k32 and k_grad_f16_k128 are built on)

// ------------------------------------------------------------------------------------------------
// k_grad_bf16_v7 (K = 64, M % 128 == 0, N % 256 == 0): v5's producer / consumer split, deeper pipeline, fewer joules.
//
// Profiling v5 (PMX_K1_PROF) showed the PRODUCERS' slot as the critical path: a serial chain of "wait for the Y tile,
// read it into the accumulator, request the next loads, 24 dependent MFMAs, ~100 VALU instructions of R -> two bf16
// terms", 3500 cycles of which the matrix pipe works 770.  Pipeline (slot s):
//   producers  GEMM1 of block s into one accumulator (starting at zero: Y is subtracted in the epilogue, so the MFMA
//              chain does not wait for it) while the epilogue of block s-1 -- VALU and LDS writes on the OTHER
//              accumulator -- sits in the same basic block and is interleaved with the MFMAs by the scheduler;
//   consumers  both gradient contractions of block s-2 from R[s & 1].
// And, because K1 runs at the package power cap (removing stall cycles returns only part of them as
// time, removing work returns all of it), less work per slot than v5:
//   * the S terms of the region's 256 columns are split ONCE and stay in LDS for the whole launch (8 blocks x 3 terms =
//     96 KB); v5 re-splits the same 32 x 64 block from fp32 on every row panel (a load, ~100 VALU instructions and six
//     LDS writes per wave and slot);
//   * Y goes from HBM straight into registers in the accumulator's layout (16 dword loads per lane and block, two rows
//     of 128 B per instruction, two blocks in flight per wave: the set block s-1 has just left is reloaded with block
//     s+1) instead of an LDS-DMA landing tile that is written and read back (32 KB of LDS traffic per slot and CU).
//     With no inline-asm requests left in the kernel every s_waitcnt vmcnt is the compiler's own exact count
//     (vmcnt(31..16) inside the slot: the older Y set has landed, the younger is in flight).
// The A terms of a new row panel are split at the top of its first slot (its rows were requested 8 slots earlier).
// LDS: Sl 96 KB + Aimg 32 KB + R 32 KB = the full 160 KB of the CU.
// HASW: weighted likelihood (nmf.py:13-41): the weights of a block travel like its Y values (a second pair of register
// sets), loss = 1/2 sum W R^2 and D = W R is what gets split and parked for the gradient contractions.
// (An intermediate build, "v6", had the pipeline but kept v5's per-panel S staging and LDS-DMA tiles: 0.427 ms where
// this one takes 0.394 and v5 0.457, same box, inside the iteration chain at 16384 x 16384.)
// ------------------------------------------------------------------------------------------------
constexpr int V7_OFF_A = V5_NB * V5_SL_BYTES, V7_OFF_R = V7_OFF_A + V5_AIMG_BYTES, V7_LDS_BYTES = V7_OFF_R + 2 * V5_R_BYTES;
static_assert(V7_OFF_R % 256 == 0, "R images must start on a bank row");
static_assert(V7_LDS_BYTES <= 160 * 1024, "");

template <bool PROF, bool HASW, bool CHAIN>
__global__ __launch_bounds__(V5_THREADS, 2) void k_grad_bf16_v7(GradV4Args a) {
    constexpr int K = 64, ROWB = 128, NCB = V5_NB;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lq = lane >> 4;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    int chainId = 0, chainPos = 0;           // CHAIN: gA summed in place along chains of workgroups, as in k_grad_f16_v8<.., CHAIN> (see there)
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if constexpr (CHAIN) {
            chain_region_map(lin, a.chainL, gx, chainId, chainPos, rowRegion, colRegion);
        } else if (gy % 8 == 0) {
            const int xcd = lin & 7, idx = lin >> 3;
            rowRegion = idx % gx;
            colRegion = xcd * (gy >> 3) + idx / gx;
        } else {
            rowRegion = lin % gx;
            colRegion = lin / gx;
        }
    }
    const int row0 = rowRegion * a.RP * V5_BM;
    const int col0 = colRegion * NCB * V5_BN;          // N % 256 == 0: every region has all 8 column blocks
    int nrp = (M - row0 + V5_BM - 1) / V5_BM;
    if (nrp > a.RP) nrp = a.RP;
    if (nrp < 0) nrp = 0;
    const int T = nrp * NCB;                 // blocks of this region (even); slots = T + 2
    const bool producer = w < 4;          // (the "no Y traffic" ablation switch of the older variants is not implemented here)
    const int j = w & 3;                     // index within the role
    float lossAcc = 0.f;
    unsigned long long ph[PROF ? 10 : 1] = {};
    const bool prof = PROF && a.prof != nullptr && (w == 0 || w == 4);
#define PH(i) if constexpr (PROF) { if (prof) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tprev; tprev = t_; } }
    unsigned long long tprev = prof ? __builtin_readcyclecounter() : 0;
    auto panel_at = [&](int t) {             // CHAIN: panel (t - chainPos) mod RP in the t-th place
        if constexpr (CHAIN) { const int p = t - chainPos; return p < 0 ? p + nrp : p; }
        else return t;
    };

    if (T <= 0) {                              // region outside the matrix: its gSt slab part and loss partial are zero
        if (!producer && (j >> 1) == 0) {      // (one gSt slab per row region)
            const int kk = (j & 1) * 32 + l31;
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
            for (int c = 0; c < NCB; ++c)
                for (int i = 0; i < 16; ++i) {
                    const int gn = col0 + c * V5_BN + tile_row(i, lane);
                    if (gn < N && a.doS) dst[(int64_t)gn * K + kk] = 0.f;
                }
        }
        if (tid == 0) a.lossPart[blockIdx.x] = 0.0;
        return;
    }

    {   // ---- all S terms of the region, once: block cb -> Sl[cb] (all 512 threads, one float4 of each block) -------
        const float4* ssrc = reinterpret_cast<const float4*>(a.St + (int64_t)col0 * K) + tid;
        const int st_off = (tid >> 4) * ROWB + (((((tid & 15) >> 1) ^ v3_swz(tid >> 4)) & 7) << 4) + 8 * (tid & 1);
        float4 sr[NCB];
#pragma unroll
        for (int c = 0; c < NCB; ++c) sr[c] = ssrc[c * (V5_BN * K / 4)];
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            bf16x4 t0, t1, t2;
            v4_split3(sr[c], t0, t1, t2);
            unsigned char* d = smem + c * V5_SL_BYTES + st_off;
            *reinterpret_cast<bf16x4*>(d) = t0;
            *reinterpret_cast<bf16x4*>(d + V5_S_TERM) = t1;
            *reinterpret_cast<bf16x4*>(d + 2 * V5_S_TERM) = t2;
        }
    }

    if (producer) {
        // ================================ producers: GEMM1 and R =================================================
        f32x16 p0, p1;
        float yE[16], yO[16];                // Y of the even / odd blocks in flight (accumulator layout)
        float wE[HASW ? 16 : 1], wO[HASW ? 16 : 1];   // their weights
        float4 areg[4][2];
        bf16x8 afr[4][3];
        // Y(b): wave-uniform base (scalar registers) + one per-lane offset; row i of the tile is a multiple of ldY further
        const int jw = __builtin_amdgcn_readfirstlane(j);
        const float* ybase0 = a.Y + (int64_t)(row0 + jw * 32) * a.ldY + col0;
        const unsigned ylane = (unsigned)(4 * hi) * (unsigned)a.ldY + (unsigned)l31;
        auto load_Y = [&](int b, float (&y)[16]) {     // block b, clamped past the end of the region
            int brp = b >> 3;
            if (brp >= nrp) brp = nrp - 1;
            brp = panel_at(brp);
            const float* base = ybase0 + (int64_t)brp * V5_BM * a.ldY + (b & 7) * V5_BN;
#pragma unroll
            // nontemporal: Y is read once per launch; keeping it out of L2 / MALL leaves the gradient slabs this kernel
            // writes there for the update kernel that folds them (iteration -2.7 % at 16384 x 16384)
            for (int i = 0; i < 16; ++i) y[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldY + ylane]);
        };
        const float* wbase0 = HASW ? a.W + (int64_t)(row0 + jw * 32) * a.ldW + col0 : nullptr;
        const unsigned wlane = HASW ? (unsigned)(4 * hi) * (unsigned)a.ldW + (unsigned)l31 : 0u;
        auto load_W = [&](int b, float (&wv)[HASW ? 16 : 1]) {
            if constexpr (HASW) {
                int brp = b >> 3;
                if (brp >= nrp) brp = nrp - 1;
                brp = panel_at(brp);
                const float* base = wbase0 + (int64_t)brp * V5_BM * a.ldW + (b & 7) * V5_BN;
#pragma unroll
                for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldW + wlane]);
            }
        };
        auto load_A = [&](int prow) {
            const float4* src = reinterpret_cast<const float4*>(a.A + (int64_t)(prow + j * 32 + l31) * K + hi * 8);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                areg[ks][0] = src[ks * 4];
                areg[ks][1] = src[ks * 4 + 1];
            }
        };
        auto make_afr = [&]() {              // split the panel rows into bf16 terms (register fragments of GEMM1's A operand)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float x[8] = {areg[ks][0].x, areg[ks][0].y, areg[ks][0].z, areg[ks][0].w,
                                    areg[ks][1].x, areg[ks][1].y, areg[ks][1].z, areg[ks][1].w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const __bf16 t0 = (__bf16)x[q];
                    const float e1 = x[q] - (float)t0;
                    const __bf16 t1 = (__bf16)e1;
                    afr[ks][0][q] = t0;
                    afr[ks][1][q] = t1;
                    afr[ks][2][q] = (__bf16)(e1 - (float)t1);
                }
            }
        };
        auto publish_A = [&]() {             // terms 0,1 of the current panel -> Aimg, for the consumers' gSt contraction
            const int pa = (j * 32 + l31) * ROWB + ((hi ^ v3_swz(j * 32 + l31)) << 4);   // chunk 2 ks + hi: ^ (ks << 5)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                *reinterpret_cast<bf16x8*>(smem + V7_OFF_A + (pa ^ (ks << 5))) = afr[ks][0];
                *reinterpret_cast<bf16x8*>(smem + V7_OFF_A + V5_A_TERM + (pa ^ (ks << 5))) = afr[ks][1];
            }
        };
        const int s_g1 = l31 * ROWB + ((hi ^ v3_swz(l31)) << 4);                 // GEMM1 B operand: row l31, chunk 2 ks + hi: ^ (ks << 5)
        const int r_w = l31 * 256 + (((4 * j) ^ v4_swz(l31)) << 4) + 8 * hi;      // R producer, ^ (g << 4)
        load_A(row0 + panel_at(0) * V5_BM);
        load_Y(0, yE);                       // slot s requests Y(s + 1) into the set block s - 1 has just left
        load_Y(1, yO);
        load_W(0, wE);
        load_W(1, wO);
        make_afr();
        if (nrp > 1) load_A(row0 + panel_at(1) * V5_BM);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();        // Sl published

        // One slot.  GEMM: block s into pc.  EPI: block s-1 from pp and its Y tile -> R[(s-1) & 1].
        auto slot = [&](int s, f32x16& pc, f32x16& pp, float (&y)[16], float (&wv)[HASW ? 16 : 1], auto gemm_c, auto epi_c) {
            constexpr bool GEMM = decltype(gemm_c)::value, EPI = decltype(epi_c)::value;
            const int cb = s & 7, rp = s >> 3;       // block s = (rp, cb); NCB == 8
            if (cb == 2 && rp < nrp) {                   // block s-2 opened this row panel: the consumers start on it in this slot
                publish_A();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
            PH(5)
            if constexpr (GEMM) {
                if (cb == 0 && s > 0) {      // block s opens a row panel: its A terms (rows requested 8 slots ago)
                    make_afr();
                    if (rp + 1 < nrp) load_A(row0 + panel_at(rp + 1) * V5_BM);
                }
            }
            bf16x8 sv[4][3];
            if constexpr (GEMM) {
                const unsigned char* Slb = smem + cb * V5_SL_BYTES;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int so = s_g1 ^ (ks << 5);
                    sv[ks][0] = *reinterpret_cast<const bf16x8*>(Slb + so);
                    sv[ks][1] = *reinterpret_cast<const bf16x8*>(Slb + so + V5_S_TERM);
                    sv[ks][2] = *reinterpret_cast<const bf16x8*>(Slb + so + 2 * V5_S_TERM);
                }
            }
            PH(2)
            if constexpr (GEMM) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pc[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][2], sv[ks][0], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], sv[ks][1], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], sv[ks][2], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], sv[ks][0], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], sv[ks][1], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], sv[ks][0], pc, 0, 0, 0);
                }
            }
            if constexpr (EPI) {
                unsigned char* Rb = smem + V7_OFF_R + ((s - 1) & 1) * V5_R_BYTES;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 h, l;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r = pp[4 * g + q] - y[4 * g + q];
                        if constexpr (HASW) {
                            const float ww = wv[4 * g + q];
                            lossAcc += ww * (r * r);
                            r *= ww;
                        } else {
                            lossAcc += r * r;
                        }
                        const __bf16 hh = (__bf16)r;
                        h[q] = hh;
                        l[q] = (__bf16)(r - (float)hh);
                    }
                    const int o = r_w ^ (g << 4);
                    *reinterpret_cast<bf16x4*>(Rb + o) = h;
                    *reinterpret_cast<bf16x4*>(Rb + V5_R_TERM + o) = l;
                }
                load_Y(s + 1, y);            // the set is free again: Y of the block two slots on
                load_W(s + 1, wv);
            }
            PH(3)
            PH(4)
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            __builtin_amdgcn_s_barrier();
            PH(0)
        };
        using yes = std::integral_constant<bool, true>;
        using no = std::integral_constant<bool, false>;
        slot(0, p0, p1, yO, wO, yes{}, no{});
#pragma nounroll
        for (int s = 1; s + 1 < T; s += 2) {
            slot(s, p1, p0, yE, wE, yes{}, yes{});
            slot(s + 1, p0, p1, yO, wO, yes{}, yes{});
        }
        slot(T - 1, p1, p0, yE, wE, yes{}, yes{});
        slot(T, p0, p1, yO, wO, no{}, yes{});
        slot(T + 1, p1, p0, yE, wE, no{}, no{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (pairs with the consumers' barrier between parking and merging their gSt row halves)
    } else {
        // ================================ consumers: GEMM2 and GEMM3 of block s-2 =================================
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // Sl published

        f32x16 accS[NCB];
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) accS[c][i] = 0.f;
        f32x16 accA0, accA1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { accA0[i] = 0.f; accA1[i] = 0.f; }
        const int kt = j & 1, mh = j >> 1;   // GEMM3 tile; GEMM2: rows 32j.., both k tiles
        int r_t0, r_t1;                      // GEMM2 A operand (R, transposing read)
        {
            const int m = j * 32 + 16 * (lq & 1) + 4 * (li & 3);
            const int n0 = 8 * hi + (li >> 2), n1 = n0 + 4;
            r_t0 = n0 * 256 + ((((m >> 3) ^ v4_swz(n0)) & 15) << 4) + 8 * ((m >> 2) & 1);
            r_t1 = n1 * 256 + ((((m >> 3) ^ v4_swz(n1)) & 15) << 4) + 8 * ((m >> 2) & 1);
        }
        auto tr_src = [&](int row, int k0) {
            const int kk = k0 + 16 * (lq & 1) + 4 * (li & 3);
            return row * ROWB + ((((kk >> 3) ^ v3_swz(row)) & 7) << 4) + 8 * ((kk >> 2) & 1);
        };
        const int s_t0 = tr_src(8 * hi + (li >> 2), 0), s_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);   // GEMM2 B operand; k tile 1: ^ 64
        const int r_g3 = l31 * 256 + (((8 * mh + hi) ^ v4_swz(l31)) << 4);                     // GEMM3 A operand, ^ (ks << 5)
        const int a_t0 = tr_src(64 * mh + 8 * hi + (li >> 2), kt * 32), a_t1 = tr_src(64 * mh + 8 * hi + 4 + (li >> 2), kt * 32);   // GEMM3 B operand
        const int slabIdxA = CHAIN ? colRegion / a.chainL : colRegion;
        auto gA_tile = [&](int prow) { return a.slabA + (int64_t)slabIdxA * M * K + (int64_t)(prow + j * 32 + 4 * hi) * K + l31; };
        auto flush_gA = [&](int prow) {
            float* p0_ = gA_tile(prow);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float* ph_ = p0_ + half * 16 * K;
                asm volatile("" : "+v"(ph_));          // keep it ONE pointer: the offsets below fold into the store's immediate
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = half * 8 + q;        // tile_row(i) = (i & 3) + 8 * (i >> 2) + 4 * hi
                    const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                    ph_[ro] = accA0[i];
                    ph_[ro + 32] = accA1[i];
                }
            }
        };
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            PH(9)
        };
        ChainLink link;                      // chain_link.h: the hand-off protocol
        if constexpr (CHAIN) link.init(a.chainFlags, chainId, nrp, j, a.status, a.wstatus, lane);
        if constexpr (CHAIN) {
            if (a.chainInject && blockIdx.x == 0 && j == 0) link.fault(3);
        }
        auto consume = [&](int b, int prow, int cb, f32x16& accSc) {     // block b: column block cb of the panel at row prow
            const unsigned char* Rb = smem + V7_OFF_R + (b & 1) * V5_R_BYTES;
            const unsigned char* Slb = smem + cb * V5_SL_BYTES;
            const unsigned char* Ab = smem + V7_OFF_A;
            if (a.doA & 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 r0 = v3_tr_pair(Rb, r_t0 + ks * 4096, r_t1 + ks * 4096);
                    const bf16x8 r1 = v3_tr_pair(Rb + V5_R_TERM, r_t0 + ks * 4096, r_t1 + ks * 4096);
                    const int so0 = s_t0 + ks * 16 * ROWB, so1 = s_t1 + ks * 16 * ROWB;
                    const bf16x8 s00 = v3_tr_pair(Slb, so0, so1);
                    const bf16x8 s01 = v3_tr_pair(Slb + V5_S_TERM, so0, so1);
                    const bf16x8 s10 = v3_tr_pair(Slb, so0 ^ 64, so1 ^ 64);
                    const bf16x8 s11 = v3_tr_pair(Slb + V5_S_TERM, so0 ^ 64, so1 ^ 64);
                    accA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, s00, accA0, 0, 0, 0);
                    accA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, s10, accA1, 0, 0, 0);
                    accA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s01, accA0, 0, 0, 0);
                    accA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s11, accA1, 0, 0, 0);
                    accA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s00, accA0, 0, 0, 0);
                    accA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, s10, accA1, 0, 0, 0);
                }
            }
            PH(6)
            if (a.doS) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);
                    const bf16x8 r0 = *reinterpret_cast<const bf16x8*>(Rb + ro);
                    const bf16x8 r1 = *reinterpret_cast<const bf16x8*>(Rb + V5_R_TERM + ro);
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
                    const bf16x8 a0 = v3_tr_pair(Ab, ao0, ao1);
                    const bf16x8 a1 = v3_tr_pair(Ab + V5_A_TERM, ao0, ao1);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r1, a0, accSc, 0, 0, 0);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, a1, accSc, 0, 0, 0);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r0, a0, accSc, 0, 0, 0);
                }
            }
            if (!CHAIN && (a.doA & 1) && cb + 1 == NCB) {
                flush_gA(prow);
#pragma unroll
                for (int i = 0; i < 16; ++i) { accA0[i] = 0.f; accA1[i] = 0.f; }
            }
        };
        sync();
        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int pnl = panel_at(rp);
            const int prow = row0 + pnl * V5_BM;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (cb == 0) {               // block s-2 opens a row panel: the producers publish its A terms now
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    if constexpr (CHAIN) link.open(pnl, chainPos, a.chainL, 1, nrp, a.chainBase, (a.doA & 1) != 0);
                }
                float pv0[4], pv1[4];
                if constexpr (CHAIN) {
                    if (cb == 3) link.look();
                    if (cb >= 4 && link.cadd) {   // piece cb - 4 of the previous sum: accumulator registers 4 (cb - 4) ..
                        const float* pb = gA_tile(prow) + 8 * (cb - 4) * K;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            pv0[q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + q * K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            pv1[q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + q * K + 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        }
                    }
                }
                consume(s - 2, prow, cb, accS[cb]);
                if constexpr (CHAIN) {
                    if (cb == 3) link.wait();   // the predecessor finished this panel about a panel-time ago: normally no spin
                    if (cb >= 4 && link.cadd) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            accA0[4 * (cb - 4) + q] += pv0[q];
                            accA1[4 * (cb - 4) + q] += pv1[q];
                        }
                    }
                    if (cb + 1 == NCB && (a.doA & 1)) {
                        flush_gA(prow);
#pragma unroll
                        for (int i = 0; i < 16; ++i) { accA0[i] = 0.f; accA1[i] = 0.f; }
                        link.flushed();
                    }
                }
                PH(7)
                sync();
                ++s;
            }
        }
        if constexpr (CHAIN) link.publish();
        // [r4] ONE gSt slab per row region: the two row halves of a tile are summed through LDS (k_grad_f16_v8.hip has the comment)
        {
            constexpr int CB = NCB / 2;
            float* fsm = reinterpret_cast<float*>(smem);
            auto halves = [&](auto MH) {
                constexpr int m = decltype(MH)::value;
                float* park = fsm + ((kt * 2 + m) * CB) * 1024 + lane;
                if (a.doS) {
#pragma unroll
                    for (int cc = 0; cc < CB; ++cc)
#pragma unroll
                        for (int i = 0; i < 16; ++i) park[cc * 1024 + i * 64] = accS[(1 - m) * CB + cc][i];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                if (a.doS) {
                    const float* oth = fsm + ((kt * 2 + (1 - m)) * CB) * 1024 + lane;
                    float* dst = a.slabS + (int64_t)rowRegion * N * K;
                    const int kk = kt * 32 + l31;
#pragma unroll
                    for (int cc = 0; cc < CB; ++cc) {
                        const int bcol = col0 + (m * CB + cc) * V5_BN;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float o = oth[cc * 1024 + i * 64], own = accS[m * CB + cc][i];
                            dst[(int64_t)(bcol + tile_row(i, lane)) * K + kk] = m == 0 ? own + o : o + own;      // rows 0-63 + rows 64-127
                        }
                    }
                }
            };
            if (mh == 0) halves(std::integral_constant<int, 0>{});
            else halves(std::integral_constant<int, 1>{});
        }
    }
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[w] = v;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < 4; ++i) s += (double)red[i];
            a.lossPart[blockIdx.x] = s;
        }
    }
    if constexpr (PROF) {
        if (prof && lane == 0)
            for (int i = 0; i < 10; ++i) atomicAdd(&a.prof[i], ph[i]);
    }
#undef PH
}

template <bool PROF, bool HASW, bool CHAIN>
static hipError_t grad_launch_bf16_v7_t(const GradV4Args& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_bf16_v7<PROF, HASW, CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_bf16_v7<PROF, HASW, CHAIN>), dim3(a.gridX * a.gridY), dim3(V5_THREADS), V7_LDS_BYTES, stream, a);
    return hipGetLastError();
}
static hipError_t grad_launch_bf16_v7(const GradV4Args& a, hipStream_t stream) {
    if (a.chainL > 0) return a.W != nullptr ? grad_launch_bf16_v7_t<false, true, true>(a, stream) : grad_launch_bf16_v7_t<false, false, true>(a, stream);
    if (a.W != nullptr) return grad_launch_bf16_v7_t<false, true, false>(a, stream);   // (no phase profiling of the weighted instance)
    return a.prof ? grad_launch_bf16_v7_t<true, false, false>(a, stream) : grad_launch_bf16_v7_t<false, false, false>(a, stream);
}

// k_grad_f16_v8 (two-term fp16 split, the shipped K = 64 kernel of mode f16x2) lives in its own file
#include "k_grad_f16_v8.hip"


template <int KP>
static size_t pipe_lds_bytes() {
    constexpr int NW = BG_THREADS / 64;
    constexpr int SL_BYTES = 3 * BG_BN * (KP + 8) * 2, STL_BYTES = 2 * KP * (BG_BN + 8) * 2;
    constexpr int NI_SL = (SL_BYTES + 1024 * NW - 1) / (1024 * NW), NI_STL = (STL_BYTES + 1024 * NW - 1) / (1024 * NW);
    return (size_t)(NI_SL + NI_STL) * NW * 1024 + (size_t)2 * KP * (BG_BM + 8) * 2 + sizeof(float) * ((size_t)BG_BM * (BG_BN + 4) + NW * 1024);
}
template <int KP, bool PROF>
static hipError_t grad_launch_bf16_pipe_t(const GradPlan& p, const GradBfArgs& a, hipStream_t stream) {
    const size_t lds = pipe_lds_bytes<KP>();
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_bf16_pipe<KP, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_bf16_pipe<KP, PROF>), dim3(p.gridX * p.gridY), dim3(BG_THREADS), lds, stream, a);
    return hipGetLastError();
}
template <int KP>
static hipError_t grad_launch_bf16_pipe(const GradPlan& p, const GradBfArgs& a, hipStream_t stream) {
    return a.prof ? grad_launch_bf16_pipe_t<KP, true>(p, a, stream) : grad_launch_bf16_pipe_t<KP, false>(p, a, stream);
}

// host side -----------------------------------------------------------------------------------------
GradPlan grad_plan_bf16(int64_t M, int64_t N, int64_t K) {
    GradPlan p{};
    p.KP = K <= 32 ? 32 : 64;
    p.BN = BG_BN;
    // PMX_K1_VARIANT (read per context; tuning A/B and the variant tests): 0 guarded kernel only, 1 LDS-DMA pipeline
    // 128 x 64 / 8 waves, 7 (default) fp32 operands split in-kernel by producer / consumer wavefronts with resident S terms
    // and Y loaded straight into registers (K = 64, M % 128 == 0, N % 256 == 0; other shapes as 1).  (4 and 5, the
    // intermediate variants of rounds 1-2, were removed in round 4: values in between select like 1.)
    p.variant = getenv("PMX_K1_VARIANT") ? atoi(getenv("PMX_K1_VARIANT")) : 7;
    const int splitA = p.KP == 64 ? 1 : 2, splitS = p.KP == 64 ? 2 : 4;
    const int64_t panels = (M + BG_BM - 1) / BG_BM;
    p.gridY = (int)((N + (int64_t)BG_CB * BG_BN - 1) / ((int64_t)BG_CB * BG_BN));
    // one resident workgroup per CU for the 8-wave variants: a single round of 256 measured 1-12 % faster than two of
    // 512 (fewer prologues / accumulator flushes, half the gSt slabs); PMX_K1_WGS overrides (tuning)
    const int wantWG = getenv("PMX_K1_WGS") ? atoi(getenv("PMX_K1_WGS")) : (p.variant >= 4 && K == 64 ? 256 : 512);
    plan_row_regions(panels, p.gridY, wantWG, &p.RP, &p.gridX);
    p.nSlabA = p.gridY * splitA;
    p.nSlabS = p.gridX * splitS;
    // [r4] k_grad_bf16_v7 / k_grad_f16_v8 (what grad_launch_bf16 runs at these shapes) merge the two row halves of gSt inside the launch
    if (p.variant >= 7 && p.KP == 64 && K == 64 && (M % V4_BM) == 0 && (N % V4_BN) == 0 && (N % (V5_NB * V5_BN)) == 0) p.nSlabS = p.gridX;
    p.ldsBytes = 2 * ((size_t)3 * BG_BN * (p.KP + 8) + (size_t)2 * p.KP * (BG_BN + 8) + (size_t)2 * p.KP * (BG_BM + 8)) +
                 sizeof(float) * ((size_t)BG_BM * (BG_BN + 4) + (size_t)(BG_THREADS / 64) * 1024);
    return p;
}

// Chained in-place accumulation of gA (k_grad_f16_v8<.., CHAIN>): members per chain for this plan, or 0 when the mode
// does not apply.  Needs: every row region with all its RP panels (the rotation), chains that are whole multiples of 8
// in number (one XCD each under round-robin dispatch), all workgroups co-resident (one per CU).
int grad_chain_length(const GradPlan& p, int64_t M, int num_cus, int longest) {
    // longest: 32 -- the whole XCD in one chain at 16384 x 16384 -- for the split-bf16 and exact-fp32 kernels; 16 for
    // k_grad_f16_v8 (round 3, PMC per chain length in profiles/r03_d_chain_length_traffic.txt: K1 writes 82 MB instead of 140 and
    // moves 1.13 x instead of 1.17 x its algorithmic bytes at the same speed -- 2486 against 2480 it/s, alternating on one box;
    // split-bf16 loses 1 % at 16, exact fp32 is indifferent).
    // PMX_K1_CHAIN: 0 switches the mode off, n >= 2 caps the chain length (tests, A/B)
    const int cap = getenv("PMX_K1_CHAIN") ? atoi(getenv("PMX_K1_CHAIN")) : longest;
    if (cap < 2) return 0;
    const int64_t panels = (M + V5_BM - 1) / V5_BM;
    if (M % V5_BM != 0 || panels % p.RP != 0) return 0;
    if (p.gridX * p.gridY > num_cus || (p.gridX * p.gridY) % 8 != 0) return 0;
    for (int L = std::min(std::min(std::min(p.RP, p.gridY), 32), cap); L >= 2; --L)
        if (p.gridY % L == 0 && ((p.gridX * p.gridY / L) % 8) == 0) return L;
    return 0;
}

template <int KP, bool EDGE>
static hipError_t grad_launch_bf16_t(const GradPlan& p, const GradBfArgs& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_bf16<KP, EDGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.ldsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_bf16<KP, EDGE>), dim3(p.gridX * p.gridY), dim3(BG_THREADS), p.ldsBytes, stream, a);
    return hipGetLastError();
}

// true when the split-bf16 path of this shape accepts a weighted likelihood (k_grad_bf16_v7<.., HASW = true>)
bool grad_bf16_takes_weights(const GradPlan& p, int64_t M, int64_t N, int64_t K);
// true when the launch below will take the variant that reads A and St as fp32 (no presplit pass needed)
bool grad_bf16_reads_fp32(const GradPlan& p, int64_t M, int64_t N, int64_t K) {
    return p.variant >= 7 && p.KP == 64 && K == 64 && (M % V4_BM) == 0 && (N % (V5_NB * V5_BN)) == 0;
}
bool grad_bf16_takes_weights(const GradPlan& p, int64_t M, int64_t N, int64_t K) {
    return grad_bf16_reads_fp32(p, M, N, K);
}
// took_f16 (optional): whether the two-term fp16 kernel is what runs -- a context in mode f16x2 drops to the split-bf16 kernel of
// the same frame at launch time when Y (or W) cannot be fetched eight bytes at a time (odd pitch, misaligned base, ldW != ldY)
hipError_t grad_launch_bf16(const GradPlan& p, const GradBfArgs& a_, const float* A, const float* St, hipStream_t stream, int* nloss, bool* took_f16 = nullptr) {
    GradBfArgs a = a_;
    a.gridX = p.gridX;
    a.gridY = p.gridY;
    *nloss = p.gridX * p.gridY;
    const int variant = p.variant;
    if (grad_bf16_reads_fp32(p, a.M, a.N, a.K)) {
        GradV4Args g{};
        g.Y = a.Y; g.ldY = a.ldY; g.A = A; g.St = St;
        g.slabA = a.slabA; g.slabS = a.slabS; g.lossPart = a.lossPart; g.status = a.status;
        g.M = a.M; g.N = a.N; g.RP = a.RP; g.doA = a.doA; g.doS = a.doS;
        g.gridX = p.gridX; g.gridY = p.gridY; g.prof = a.prof;
        g.W = a.W; g.ldW = a.ldW;
        g.absmax = a.absmax; g.ymax = a.ymax; g.wmax = a.wmax;
        g.chainL = a.chainL; g.chainFlags = a.chainFlags; g.chainBase = a.chainBase; g.wstatus = a.wstatus; g.chainInject = a.chainInject; g.rangeRatio = a.rangeRatio; g.r3 = a.r3; g.consPrio = a.consPrio;
        // fp16 two-term mode; its producers fetch Y (and W) eight bytes at a time: even pitch, 8-byte-aligned base (anything
        // else runs the split-bf16 kernel of the same frame below)
        const bool pairs_ok = (a.ldY % 2) == 0 && (((uintptr_t)a.Y) & 7) == 0 && (a.W == nullptr || (a.ldW == a.ldY && (((uintptr_t)a.W) & 7) == 0));   // (the weights share Y's per-lane offsets)
        const bool f16 = a.absmax != nullptr && pairs_ok;
        if (took_f16) *took_f16 = f16;
        if (f16) return grad_launch_f16_v8(g, stream);
        return grad_launch_bf16_v7(g, stream);   // plain loads: any row pitch
    }
    if (a.W != nullptr) return hipErrorInvalidValue;
    // Whole blocks with 16-byte-aligned rows take an LDS-DMA variant; anything else the guarded kernel.
    const bool aligned = (a.ldY % 4) == 0 && (((uintptr_t)a.Y) & 15) == 0;
    const bool edge = (a.M % BG_BM) != 0 || (a.N % BG_BN) != 0 || !aligned;
    if (!edge && variant >= 1) return p.KP == 32 ? grad_launch_bf16_pipe<32>(p, a, stream) : grad_launch_bf16_pipe<64>(p, a, stream);
    if (p.KP == 32) return edge ? grad_launch_bf16_t<32, true>(p, a, stream) : grad_launch_bf16_t<32, false>(p, a, stream);
    return edge ? grad_launch_bf16_t<64, true>(p, a, stream) : grad_launch_bf16_t<64, false>(p, a, stream);
}

void launch_presplit(const PresplitArgs& a, hipStream_t s) {
    const int64_t mx = a.rowsPad[0] > a.rowsPad[1] ? a.rowsPad[0] : a.rowsPad[1];
    hipLaunchKernelGGL(k_presplit, dim3((unsigned)(mx / 64), 2), dim3(256), 0, s, a);
}
