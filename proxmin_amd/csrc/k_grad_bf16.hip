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

template <int KP>
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
    };

    if (nsteps > 0) {   // prologue: same issue order as inside the loop
        if (!noY) dma_Y(row0, col0);
        else { lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds));
               lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); lds_dma16(a.Sp, __builtin_amdgcn_readfirstlane(ytile_lds)); }
        dma_Sl(col0);
        dma_Stl(col0);
    }
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool prof = a.prof != nullptr && w == 0;
#define PH(i) if (prof) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tprev; tprev = t_; }
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
    if (prof && lane == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&a.prof[i], ph[i]);
#undef PH
}

template <int KP>
static size_t pipe_lds_bytes() {
    constexpr int NW = BG_THREADS / 64;
    constexpr int SL_BYTES = 3 * BG_BN * (KP + 8) * 2, STL_BYTES = 2 * KP * (BG_BN + 8) * 2;
    constexpr int NI_SL = (SL_BYTES + 1024 * NW - 1) / (1024 * NW), NI_STL = (STL_BYTES + 1024 * NW - 1) / (1024 * NW);
    return (size_t)(NI_SL + NI_STL) * NW * 1024 + (size_t)2 * KP * (BG_BM + 8) * 2 + sizeof(float) * ((size_t)BG_BM * (BG_BN + 4) + NW * 1024);
}
template <int KP>
static hipError_t grad_launch_bf16_pipe(const GradPlan& p, const GradBfArgs& a, hipStream_t stream) {
    const size_t lds = pipe_lds_bytes<KP>();
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_bf16_pipe<KP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_bf16_pipe<KP>), dim3(p.gridX * p.gridY), dim3(BG_THREADS), lds, stream, a);
    return hipGetLastError();
}

// host side -----------------------------------------------------------------------------------------
GradPlan grad_plan_bf16(int64_t M, int64_t N, int64_t K) {
    GradPlan p{};
    p.KP = K <= 32 ? 32 : 64;
    p.BN = BG_BN;
    const int splitA = p.KP == 64 ? 1 : 2, splitS = p.KP == 64 ? 2 : 4;
    const int64_t panels = (M + BG_BM - 1) / BG_BM;
    p.gridY = (int)((N + (int64_t)BG_CB * BG_BN - 1) / ((int64_t)BG_CB * BG_BN));
    int64_t wantX = (512 + p.gridY - 1) / p.gridY;
    if (wantX < 1) wantX = 1;
    if (wantX > panels) wantX = panels;
    p.RP = (int)((panels + wantX - 1) / wantX);
    p.gridX = (int)((panels + p.RP - 1) / p.RP);
    p.nSlabA = p.gridY * splitA;
    p.nSlabS = p.gridX * splitS;
    p.ldsBytes = 2 * ((size_t)3 * BG_BN * (p.KP + 8) + (size_t)2 * p.KP * (BG_BN + 8) + (size_t)2 * p.KP * (BG_BM + 8)) +
                 sizeof(float) * ((size_t)BG_BM * (BG_BN + 4) + (size_t)(BG_THREADS / 64) * 1024);
    return p;
}

template <int KP, bool EDGE>
static hipError_t grad_launch_bf16_t(const GradPlan& p, const GradBfArgs& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_bf16<KP, EDGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.ldsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_bf16<KP, EDGE>), dim3(p.gridX * p.gridY), dim3(BG_THREADS), p.ldsBytes, stream, a);
    return hipGetLastError();
}

hipError_t grad_launch_bf16(const GradPlan& p, const GradBfArgs& a_, hipStream_t stream) {
    GradBfArgs a = a_;
    a.gridX = p.gridX;
    a.gridY = p.gridY;
    // Whole 128 x 64 blocks with 16-byte-aligned rows take the LDS-DMA variant; anything else the guarded one.
    const bool edge = (a.M % BG_BM) != 0 || (a.N % BG_BN) != 0 || (a.ldY % 4) != 0 || (((uintptr_t)a.Y) & 15) != 0;
    static const int variant = getenv("PMX_K1_VARIANT") ? atoi(getenv("PMX_K1_VARIANT")) : 1;   // 0: un-pipelined (tuning A/B)
    if (!edge && variant == 1) return p.KP == 32 ? grad_launch_bf16_pipe<32>(p, a, stream) : grad_launch_bf16_pipe<64>(p, a, stream);
    if (p.KP == 32) return edge ? grad_launch_bf16_t<32, true>(p, a, stream) : grad_launch_bf16_t<32, false>(p, a, stream);
    return edge ? grad_launch_bf16_t<64, true>(p, a, stream) : grad_launch_bf16_t<64, false>(p, a, stream);
}

void launch_presplit(const PresplitArgs& a, hipStream_t s) {
    const int64_t mx = a.rowsPad[0] > a.rowsPad[1] ? a.rowsPad[0] : a.rowsPad[1];
    hipLaunchKernelGGL(k_presplit, dim3((unsigned)(mx / 64), 2), dim3(256), 0, s, a);
}
