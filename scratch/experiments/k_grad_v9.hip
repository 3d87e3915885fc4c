// k_grad_f16_v9 (K = 64, M % 128 == 0, N % 256 == 0): the two-term fp16 fused residual-gradient kernel with ROW-OWNING waves.
//
// k_grad_f16_v8 splits a workgroup into producer waves (A S - Y) and consumer waves (the two gradient contractions) that
// meet at a barrier every 128 x 32 block, and hands R, the A panel and the S block to the consumers through LDS.  Here a
// workgroup is four waves, one per SIMD, and wave j does ALL THREE contractions for rows 32 j .. 32 j + 31 of the panel:
//
//     P  (32 m x 32 n)  = A_j S_blk           A_j fragments in registers for the panel, S from the resident LDS image
//     R  = P - Y                               in the accumulator layout: lane = column n, registers = rows m
//     gSt (32 n x 64 k) += R^T A_j             R^T is an MFMA A-operand EXACTLY as it sits in the registers (row = lane = n);
//                                              the contraction index m arrives in the accumulator's row order, and the
//                                              B-operand (A_j, held in registers for the panel) is loaded in that same
//                                              order -- a contraction may be summed in any order.  No LDS, no shuffle.
//     gA  (32 m x 64 k) += R S_blk^T           needs R with lane = m: the wave parks its 32 x 32 tile in a wave-private
//                                              corner of the [n][m] image and reads it back transposed (ds_read_b64_tr_b16)
//
// so R never crosses waves: no barrier inside the loop (waves drift freely; the only workgroup-wide barriers are the S
// staging in the prologue and the gSt reduction in the epilogue), no A image in LDS, and a third less LDS traffic per block
// (no R / A reads for gSt).  Each wave holds the gSt accumulators of all NB column blocks for ITS rows (NB x 2 tiles); the
// four partial sums are added through LDS once, at the end of the launch: one gSt slab per row region (v8: two).
// One wave per SIMD means 512 registers per lane and no co-resident wave to hide latencies: Y is requested two blocks
// ahead, the next panel's rows of A one panel ahead (through a wave-private LDS staging area, from which the wave reads
// them back in the two layouts it needs), and the instruction stream of a block is laid out so that the matrix pipe
// always has independent work queued: [GEMM1 of block s] overlaps [epilogue + gSt + gA of block s - 1].
// Arithmetic (operand scales, split, products kept) is k_grad_f16_v8's; results agree with it to summation order.
#include "pmx_common.h"

constexpr int V9_THREADS = 256;
constexpr int V9_NB = 4;                                // column blocks (of 32) per region: 128 columns
constexpr int V9_A_STRIDE = 272;                        // bytes per staged row of A (64 fp32 + 16: conflict-free b128 / b32 reads)
constexpr int V9_A_BYTES = 128 * V9_A_STRIDE;
template <int NB> struct V9Lds {
    static constexpr int OFF_R = NB * V8_SL_BYTES;      // S image: NB blocks x 2 terms x 4 KB
    static constexpr int OFF_A = OFF_R + V5_R_BYTES;    // R image: 2 terms x 8 KB ([n][128 m], wave j owns m = 32 j ..)
    static constexpr int BYTES = OFF_A + V9_A_BYTES;    // A staging: 128 rows fp32, wave j owns rows 32 j ..
};
static_assert(V9Lds<V9_NB>::BYTES <= 160 * 1024, "");
static_assert(V9Lds<V9_NB>::OFF_R >= 4 * 2 * 16 * 64 * 4, "the S image doubles as the 32 KB scratch of the final gSt reduction");

// DOA / DOS: which gradients the launch produces (compile-time: a run-time test would cut the hand-laid instruction
// stream of a block into basic blocks the scheduler cannot interleave)
template <bool HASW, bool CHAIN, int NB, bool DOA, bool DOS>
__global__ __launch_bounds__(V9_THREADS, 1) void k_grad_f16_v9(GradV4Args a) {
    constexpr int K = 64, ROWB = 128;
    static_assert(DOA || !CHAIN, "the chains carry gA");
    using L = V9Lds<NB>;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lq = lane >> 4;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    int chainId = 0, chainPos = 0;
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if constexpr (CHAIN) {       // see k_grad_f16_v8: a chain's members are 8 apart in dispatch order (one XCD)
            const int Lc = a.chainL, xcd = lin & 7, idx = lin >> 3;
            chainPos = idx % Lc;
            chainId = (idx / Lc) * 8 + xcd;
            rowRegion = chainId % gx;
            colRegion = (chainId / gx) * Lc + chainPos;
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
    const int col0 = colRegion * NB * V5_BN;
    int nrp = (M - row0 + V5_BM - 1) / V5_BM;
    if (nrp > a.RP) nrp = a.RP;
    if (nrp < 0) nrp = 0;
    auto panel_at = [&](int t) {
        if constexpr (CHAIN) { const int p = t - chainPos; return p < 0 ? p + nrp : p; }
        else return t;
    };
    float lossAcc = 0.f;

    // ---- power-of-two operand scales (k_grad_f16_v8) ------------------------------------------------------------------
    float scA, scS, scR, unP, unA, unS;
    {
        float* red = reinterpret_cast<float*>(smem);
        float m0 = 0.f, m1 = 0.f;
        for (int i = tid; i < V8_NPART; i += V9_THREADS) { m0 = fmaxf(m0, a.absmax[i]); m1 = fmaxf(m1, a.absmax[V8_NPART + i]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
        if (lane == 0) { red[j] = m0; red[8 + j] = m1; }
        __syncthreads();
        float mA = red[0], mS = red[8];
        for (int i = 1; i < 4; ++i) { mA = fmaxf(mA, red[i]); mS = fmaxf(mS, red[8 + i]); }
        __syncthreads();
        int qA = 0, qS = 0, qR = 0;
        (void)frexpf(mA, &qA);
        (void)frexpf(mS, &qS);
        (void)frexpf((a.ymax + (float)K * mA * mS) * a.wmax, &qR);
        const int eA = mA > 0.f ? 14 - qA : 0, eS = mS > 0.f ? 14 - qS : 0, eR = 14 - qR;
        scA = ldexpf(1.f, eA); scS = ldexpf(1.f, eS); scR = ldexpf(1.f, eR);
        unP = ldexpf(1.f, -(eA + eS)); unA = ldexpf(1.f, -(eR + eS)); unS = ldexpf(1.f, -(eR + eA));
    }
    if (nrp <= 0) {                          // region outside the matrix (cannot happen with the chained plan)
        if (DOS) {
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
            for (int e = tid; e < NB * V5_BN * K; e += V9_THREADS) {
                const int gn = col0 + e / K;
                if (gn < N) dst[(int64_t)gn * K + (e % K)] = 0.f;
            }
        }
        if (tid == 0) a.lossPart[blockIdx.x] = 0.0;
        return;
    }
    {   // ---- the S terms of the region's NB blocks, once: block c -> Sl[c] (256 threads: two float4 of each block) ----
        const float4* ssrc = reinterpret_cast<const float4*>(a.St + (int64_t)col0 * K);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t2 = tid + h * V9_THREADS;
            const int st_off = (t2 >> 4) * ROWB + (((((t2 & 15) >> 1) ^ v3_swz(t2 >> 4)) & 7) << 4) + 8 * (t2 & 1);
            float4 sr[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) sr[c] = ssrc[c * (V5_BN * K / 4) + t2];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                f16x4 t0, t1;
                v8_split2(sr[c], scS, t0, t1);
                unsigned char* d = smem + c * V8_SL_BYTES + st_off;
                *reinterpret_cast<f16x4*>(d) = t0;
                *reinterpret_cast<f16x4*>(d + V5_S_TERM) = t1;
            }
        }
    }

    // ---- per-wave state ---------------------------------------------------------------------------------------------
    f32x16 accS[NB][2];                      // gSt of this wave's rows: [column block][k tile]
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accS[c][0][i] = 0.f; accS[c][1][i] = 0.f; }
    f32x16 accA0, accA1;                     // gA of the current panel's rows (k tiles 0, 1)
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA0[i] = 0.f; accA1[i] = 0.f; }
    f16x8 afr[4][2];                         // GEMM1 A operand: row l31, k = 16 ks + 8 hi + q        [ks][term]
    f16x8 bfr[2][2][2];                      // gSt B operand: column k = l31 + 32 kt, contraction slot 8 hi + t <-> row
                                             //   m = 16 mm + 4 hi + (t & 3) + 8 (t >> 2)            [kt][mm][term]
    float yE[16], yO[16];                    // Y of the even / odd blocks in flight (accumulator layout)
    float wE[HASW ? 16 : 1], wO[HASW ? 16 : 1];

    unsigned char* const Abuf = smem + L::OFF_A + j * 32 * V9_A_STRIDE;        // this wave's 32 staged rows
    const int T = nrp * NB;                                                      // blocks of this region
    const float* ybase0 = a.Y + (int64_t)(row0 + j * 32) * a.ldY + col0;
    const unsigned ylane = (unsigned)(4 * hi) * (unsigned)a.ldY + (unsigned)l31;
    auto load_Y = [&](int b, float (&y)[16]) {
        int bt = b / NB;
        if (bt >= nrp) bt = nrp - 1;
        const float* base = ybase0 + (int64_t)panel_at(bt) * V5_BM * a.ldY + (b % NB) * V5_BN;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldY + ylane]);
    };
    const float* wbase0 = HASW ? a.W + (int64_t)(row0 + j * 32) * a.ldW + col0 : nullptr;
    const unsigned wlane = HASW ? (unsigned)(4 * hi) * (unsigned)a.ldW + (unsigned)l31 : 0u;
    auto load_W = [&](int b, float (&wv)[HASW ? 16 : 1]) {
        if constexpr (HASW) {
            int bt = b / NB;
            if (bt >= nrp) bt = nrp - 1;
            const float* base = wbase0 + (int64_t)panel_at(bt) * V5_BM * a.ldW + (b % NB) * V5_BN;
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldW + wlane]);
        }
    };
    // rows of A for the panel visited in place t: global -> registers (8 float4 per lane: row l31, k = 8 hi + 16 ks ..)
    float4 areg[4][2];
    auto fetch_A = [&](int t) {
        const float4* src = reinterpret_cast<const float4*>(a.A + (int64_t)(row0 + panel_at(t) * V5_BM + j * 32 + l31) * K + hi * 8);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { areg[ks][0] = src[ks * 4]; areg[ks][1] = src[ks * 4 + 1]; }
    };
    auto stage_A = [&]() {                   // registers -> the wave's staging rows (fp32)
        unsigned char* d = Abuf + l31 * V9_A_STRIDE + hi * 32;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            *reinterpret_cast<float4*>(d + ks * 64) = areg[ks][0];
            *reinterpret_cast<float4*>(d + ks * 64 + 16) = areg[ks][1];
        }
    };
    auto make_afr = [&]() {                  // staging rows -> GEMM1's A fragments (scaled, two fp16 terms)
        const unsigned char* s = Abuf + l31 * V9_A_STRIDE + hi * 32;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 x0 = *reinterpret_cast<const float4*>(s + ks * 64), x1 = *reinterpret_cast<const float4*>(s + ks * 64 + 16);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float xs = x[q] * scA;
                const _Float16 t0 = (_Float16)xs;
                afr[ks][0][q] = t0;
                afr[ks][1][q] = (_Float16)(xs - (float)t0);
            }
        }
    };
    auto make_bfr = [&]() {                  // staging rows -> the gSt contraction's B fragments (rows in accumulator order)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int m = 16 * mm + 4 * hi + (t & 3) + 8 * (t >> 2);
                    const float xs = *reinterpret_cast<const float*>(Abuf + m * V9_A_STRIDE + (l31 + 32 * kt) * 4) * scA;
                    const _Float16 t0 = (_Float16)xs;
                    bfr[kt][mm][0][t] = t0;
                    bfr[kt][mm][1][t] = (_Float16)(xs - (float)t0);
                }
    };

    // LDS addresses (k_grad_f16_v8's images and swizzles)
    const int s_g1 = l31 * ROWB + ((hi ^ v3_swz(l31)) << 4);                        // GEMM1 B operand, ^ (ks << 5)
    const int r_w = l31 * 256 + (((4 * j) ^ v4_swz(l31)) << 4) + 8 * hi;             // R tile store, ^ (g << 4)
    int r_t0, r_t1;                                                                   // gA's A operand (R, transposing read)
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
    const int s_t0 = tr_src(8 * hi + (li >> 2), 0), s_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);      // gA's B operand; k tile 1: ^ 64
    unsigned char* const Rb = smem + L::OFF_R;

    // ---- gA hand-off (slab or chain): see k_grad_f16_v8 ---------------------------------------------------------------
    const int slabIdxA = CHAIN ? colRegion / a.chainL : colRegion;
    auto gA_tile = [&](int prow) { return a.slabA + (int64_t)slabIdxA * M * K + (int64_t)(prow + j * 32 + 4 * hi) * K + l31; };
    auto flush_gA = [&](int prow) {
        float* p0_ = gA_tile(prow);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float* ph_ = p0_ + half * 16 * K;
            asm volatile("" : "+v"(ph_));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = half * 8 + q;
                const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                ph_[ro] = accA0[i] * unA;
                ph_[ro + 32] = accA1[i] * unA;
            }
        }
    };
    const float invUnA = scR * scS;
    unsigned* cflags = nullptr;
    unsigned myxcc = 0;
    if constexpr (CHAIN) {
        cflags = a.chainFlags + (size_t)chainId * nrp * 4 + j;
        myxcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    }
    unsigned* pendFlag = nullptr;
    unsigned pendVal = 0;
    unsigned* curFlag = nullptr;
    unsigned cwant = 0, cseen = 0;
    bool cadd = false, cdead = false;
    auto chain_fault = [&](int code) {
        if (lane == 0 && code > 0) {
            a.wstatus->k1_fault = code;
            a.wstatus->reason = HALT_ERROR;
            __threadfence();
            a.wstatus->halt = 1;
        }
        cadd = false;
        cdead = true;
    };
    auto chain_publish = [&]() {
        if constexpr (CHAIN) {
            if (pendFlag != nullptr) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(pendFlag, pendVal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pendFlag = nullptr;
            }
        }
    };

    // ---- prologue: first panel's rows, first two blocks of Y ----------------------------------------------------------
    fetch_A(0);
    load_Y(0, yE);
    load_Y(1, yO);
    load_W(0, wE);
    load_W(1, wO);
    stage_A();
    __syncthreads();                         // S image published (and this wave's staged rows are its own)
    make_afr();
    make_bfr();
    if constexpr (CHAIN) {
        if (a.chainInject && blockIdx.x == 0 && j == 0) chain_fault(3);
    }

    // ---- the pieces of one block's work.  A block's three contractions depend on each other (P -> R -> gSt, gA), so the
    // stream of ONE wave is software-pipelined by hand: body(s) issues GEMM1 of block s step by step and lays the
    // epilogue, the gSt and the gA contraction of block s - 1 between those steps -- independent work for the matrix pipe
    // while the vector unit converts, and vice versa. ----------------------------------------------------------------------
    f32x16 pP;                               // P of the previous block
#pragma unroll
    for (int i = 0; i < 16; ++i) pP[i] = 0.f;
    auto gemm1_step = [&](int cb, int ks, f32x16& pc) {
        const unsigned char* Slb = smem + cb * V8_SL_BYTES;
        const int so = s_g1 ^ (ks << 5);
        const f16x8 sv0 = *reinterpret_cast<const f16x8*>(Slb + so);
        const f16x8 sv1 = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][1], sv0, pc, 0, 0, 0);
        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sv1, pc, 0, 0, 0);
        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sv0, pc, 0, 0, 0);
    };
    // residual, loss, split of registers 4 g .. 4 g + 3 of the previous block; the tile goes to the wave's corner of the
    // [n][m] image for the transposed read
    auto epi_group = [&](int g, float (&y)[16], float (&wv)[HASW ? 16 : 1], f16x8 (&rh)[2], f16x8 (&rl)[2]) {
        f16x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float r = pP[4 * g + q] * unP - y[4 * g + q];
            if constexpr (HASW) {
                const float ww = wv[4 * g + q];
                lossAcc += ww * (r * r);
                r *= ww;
            } else {
                lossAcc += r * r;
            }
            const float rs = r * scR;
            const _Float16 hh = (_Float16)rs;
            h[q] = hh;
            l[q] = (_Float16)(rs - (float)hh);
            rh[g >> 1][4 * (g & 1) + q] = h[q];
            rl[g >> 1][4 * (g & 1) + q] = l[q];
        }
        if constexpr (DOA) {
            const int o = r_w ^ (g << 4);
            *reinterpret_cast<f16x4*>(Rb + o) = h;
            *reinterpret_cast<f16x4*>(Rb + V5_R_TERM + o) = l;
        }
    };
    auto gemm3_group = [&](int mm, f32x16 (&aS)[2], const f16x8 (&rh)[2], const f16x8 (&rl)[2]) {   // gSt += R^T A: registers only
        if constexpr (DOS) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                aS[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl[mm], bfr[kt][mm][0], aS[kt], 0, 0, 0);
                aS[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[mm], bfr[kt][mm][1], aS[kt], 0, 0, 0);
                aS[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[mm], bfr[kt][mm][0], aS[kt], 0, 0, 0);
            }
        }
    };
    auto gemm2_step = [&](int cbp, int ks) {  // gA += R S^T: R back from the tile, transposed; S^T by the transposing read
        if constexpr (DOA) {
            const unsigned char* Slb = smem + cbp * V8_SL_BYTES;
            const f16x8 r0 = v8_tr_pair(Rb, r_t0 + ks * 4096, r_t1 + ks * 4096);
            const f16x8 r1 = v8_tr_pair(Rb + V5_R_TERM, r_t0 + ks * 4096, r_t1 + ks * 4096);
            const int so0 = s_t0 + ks * 16 * ROWB, so1 = s_t1 + ks * 16 * ROWB;
            const f16x8 s00 = v8_tr_pair(Slb, so0, so1);
            const f16x8 s01 = v8_tr_pair(Slb + V5_S_TERM, so0, so1);
            const f16x8 s10 = v8_tr_pair(Slb, so0 ^ 64, so1 ^ 64);
            const f16x8 s11 = v8_tr_pair(Slb + V5_S_TERM, so0 ^ 64, so1 ^ 64);
            accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s00, accA0, 0, 0, 0);
            accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s10, accA1, 0, 0, 0);
            accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s01, accA0, 0, 0, 0);
            accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s11, accA1, 0, 0, 0);
            accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s00, accA0, 0, 0, 0);
            accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s10, accA1, 0, 0, 0);
        }
    };
    // body(s): GEMM1 of block s (column block cb of the current panel) beside the rest of block s - 1 (column block cbp,
    // accumulators aSp, Y set y); CUR / PREV switch the two halves off at the ends of the region.
    auto body = [&](int s, int cb, int cbp, f32x16 (&aSp)[2], float (&y)[16], float (&wv)[HASW ? 16 : 1], auto cur_c, auto prev_c) {
        constexpr bool CUR = decltype(cur_c)::value, PREV = decltype(prev_c)::value;
        f32x16 pN;
#pragma unroll
        for (int i = 0; i < 16; ++i) pN[i] = 0.f;
        f16x8 rh[2], rl[2];
        if constexpr (CUR) gemm1_step(cb, 0, pN);
        if constexpr (PREV) { epi_group(0, y, wv, rh, rl); epi_group(1, y, wv, rh, rl); }
        if constexpr (CUR) gemm1_step(cb, 1, pN);
        if constexpr (PREV) {
            epi_group(2, y, wv, rh, rl);
            epi_group(3, y, wv, rh, rl);
            load_Y(s + 1, y);                            // the set is free again: Y of the block two places on
            load_W(s + 1, wv);
        }
        if constexpr (CUR) gemm1_step(cb, 2, pN);
        if constexpr (PREV) gemm3_group(0, aSp, rh, rl);
        if constexpr (CUR) gemm1_step(cb, 3, pN);
        if constexpr (PREV) {
            gemm3_group(1, aSp, rh, rl);
            gemm2_step(cbp, 0);
            gemm2_step(cbp, 1);
        }
        pP = pN;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    // chain state of the panel whose gA is accumulating
    int accPrw = row0 + panel_at(0) * V5_BM;
    auto chain_open = [&](int pnl) {
        if constexpr (CHAIN) {
            const int c = chainPos, Lc = a.chainL;
            const int nw = pnl + Lc - nrp > 0 ? pnl + Lc - nrp : 0;
            const int k = pnl + c >= nrp ? pnl + c - nrp : c + nw;
            cadd = DOA && k > 0 && !cdead;
            cwant = a.chainBase + (unsigned)k;
            curFlag = cflags + pnl * 4;
        }
    };
    auto chain_poll_issue = [&]() {
        if constexpr (CHAIN) { if (cadd) cseen = __hip_atomic_load(curFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    };
    auto chain_poll_wait = [&]() {
        if constexpr (CHAIN) {
            if (cadd) {
                unsigned v = __builtin_amdgcn_readfirstlane(cseen);
                if ((v >> 4) != cwant) {
                    const long long t0 = wall_clock64();
                    for (int spins = 1; (v >> 4) != cwant; ++spins) {
                        if ((spins & 63) == 0) {
                            if (chain_halted(a.status)) { chain_fault(0); break; }
                            if (wall_clock64() - t0 > 2000000) { chain_fault(1); break; }
                        }
                        __builtin_amdgcn_s_sleep(8);
                        v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(curFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
                }
                if (cadd && (v & 15u) != myxcc) chain_fault(2);
            }
        }
    };
    // previous sum of the accumulating panel, registers 8 half .. 8 half + 7 of both k tiles
    auto chain_load = [&](int half, float (&pv)[16]) {
        if constexpr (CHAIN) {
            if (cadd) {
                const float* pb = gA_tile(accPrw) + 16 * half * K;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                    pv[q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + ro), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    pv[8 + q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + ro + 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
            }
        }
    };
    auto chain_add = [&](int half, const float (&pv)[16]) {
        if constexpr (CHAIN) {
            if (cadd) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    accA0[8 * half + q] += pv[q] * invUnA;
                    accA1[8 * half + q] += pv[8 + q] * invUnA;
                }
            }
        }
    };
    auto finish_panel = [&]() {              // gA of the accumulating panel is complete
        if constexpr (DOA) {
            flush_gA(accPrw);
#pragma unroll
            for (int i = 0; i < 16; ++i) { accA0[i] = 0.f; accA1[i] = 0.f; }
            if constexpr (CHAIN) {
                pendFlag = curFlag;
                pendVal = ((cwant + 1u) << 4) | myxcc;
            }
        }
    };
    static_assert(NB == 4, "the hand-laid pipeline below is written for four column blocks per panel");
    chain_open(panel_at(0));

    int s = 0;
#pragma nounroll
    for (int t = 0; t < nrp; ++t) {
        float pv[16];
        // -- column block 0: GEMM1(t, 0) beside the last block of panel t - 1, whose gA is complete afterwards
        if (t == 0) body(s, 0, 3, accS[3], yO, wO, yes{}, no{});
        else {
            body(s, 0, 3, accS[3], yO, wO, yes{}, yes{});
            finish_panel();
            accPrw = row0 + panel_at(t) * V5_BM;
            chain_open(panel_at(t));
            make_bfr();                      // panel t's rows in the gSt contraction's order (panel t - 1's are done with)
        }
        ++s;
        // -- column block 1
        chain_publish();                     // the previous panel's arrival: its stores were issued a block ago
        if (t + 1 < nrp) fetch_A(t + 1);     // next panel's rows: requested now, parked in the staging rows one block later
        chain_poll_issue();
        body(s, 1, 0, accS[0], yE, wE, yes{}, yes{});
        chain_poll_wait();
        ++s;
        // -- column block 2
        if (t + 1 < nrp) stage_A();
        chain_load(0, pv);
        body(s, 2, 1, accS[1], yO, wO, yes{}, yes{});
        chain_add(0, pv);
        ++s;
        // -- column block 3
        chain_load(1, pv);
        body(s, 3, 2, accS[2], yE, wE, yes{}, yes{});
        chain_add(1, pv);
        ++s;
        if (t + 1 < nrp) make_afr();         // GEMM1 of panel t is done: the next panel's fragments
    }
    body(s, 0, 3, accS[3], yO, wO, no{}, yes{});      // drain: the last block's epilogue and contractions
    finish_panel();
    chain_publish();
    (void)T;

    // ---- gSt: the four waves' partial sums, added through LDS in a fixed order, one slab per row region ----------------
    if constexpr (DOS) {
        float* F = reinterpret_cast<float*>(smem);             // [wave][kt][i][lane]
        float* dst = a.slabS + (int64_t)rowRegion * N * K;
        const int kt_o = j & 1, ih = j >> 1;                    // this wave adds tile kt_o, registers 8 ih .. 8 ih + 7
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            __syncthreads();                                    // (first round: every wave is done with the S image)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) F[((j * 2 + kt) * 16 + i) * 64 + lane] = accS[c][kt][i];
            __syncthreads();
            const int bcol = col0 + c * V5_BN;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = 8 * ih + q;
                float v = F[((0 * 2 + kt_o) * 16 + i) * 64 + lane];
                v += F[((1 * 2 + kt_o) * 16 + i) * 64 + lane];
                v += F[((2 * 2 + kt_o) * 16 + i) * 64 + lane];
                v += F[((3 * 2 + kt_o) * 16 + i) * 64 + lane];
                dst[(int64_t)(bcol + tile_row(i, lane)) * K + kt_o * 32 + l31] = v * unS;
            }
        }
    }
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[j] = v;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int i = 0; i < 4; ++i) sum += (double)red[i];
            a.lossPart[blockIdx.x] = sum;
        }
    }
}

template <bool HASW, bool CHAIN, bool DOA, bool DOS>
static hipError_t grad_launch_f16_v9_t(const GradV4Args& a, hipStream_t stream) {
    constexpr int NB = V9_NB;
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f16_v9<HASW, CHAIN, NB, DOA, DOS>, hipFuncAttributeMaxDynamicSharedMemorySize, V9Lds<NB>::BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f16_v9<HASW, CHAIN, NB, DOA, DOS>), dim3(a.gridX * a.gridY), dim3(V9_THREADS), V9Lds<NB>::BYTES, stream, a);
    return hipGetLastError();
}
template <bool HASW>
static hipError_t grad_launch_f16_v9_w(const GradV4Args& a, hipStream_t stream) {
    const bool dA = (a.doA & 1) != 0, dS = a.doS != 0, ch = a.chainL > 0;
    if (dA && dS) return ch ? grad_launch_f16_v9_t<HASW, true, true, true>(a, stream) : grad_launch_f16_v9_t<HASW, false, true, true>(a, stream);
    if (dA) return ch ? grad_launch_f16_v9_t<HASW, true, true, false>(a, stream) : grad_launch_f16_v9_t<HASW, false, true, false>(a, stream);
    if (dS) return grad_launch_f16_v9_t<HASW, false, false, true>(a, stream);
    return grad_launch_f16_v9_t<HASW, false, false, false>(a, stream);
}
static hipError_t grad_launch_f16_v9(const GradV4Args& a, hipStream_t stream) {
    return a.W != nullptr ? grad_launch_f16_v9_w<true>(a, stream) : grad_launch_f16_v9_w<false>(a, stream);
}
