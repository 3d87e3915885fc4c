// K1 in exact fp32 (v_mfma_f32_32x32x2_f32) with producer / consumer wavefronts: k_grad_f32_pc<K, HASW>, K = 32 or 64,
// M % 128 = 0, N % 256 = 0 (included by pmx_api.hip after k_grad_bf16.hip; restates nmf.py:25,39-41 like every K1).
//
// k_grad_f32 (k_grad.hip) walks all 8 waves of a workgroup through load -> GEMM1 -> barrier -> GEMM2 -> GEMM3 phases
// and keeps the fp32 matrix pipe 65 % busy (42 % on the short grids of 4096 x 4096).  The fp32 MFMA is SLOW -- 64 cycles per
// 32 x 32 x 2 instruction, 6144 matrix-pipe cycles per SIMD for one 128 x 32 block at K = 64 -- so everything else (operand
// fetches, the residual, barriers) fits in its shadow if something is always there to issue it.  This kernel is the
// frame of k_grad_f16_v8 with fp32 operands: four producer waves (P = A S for their 32 rows of the 128 x 32 block, R = P - Y
// parked in LDS), four consumer waves two blocks behind (gA = R S^T, gSt = R^T A), one producer and one consumer per
// SIMD, one barrier per block, the S rows of the workgroup's 256 columns resident in LDS for the whole launch, A's rows in
// the producers' registers for a panel (next panel prefetched), gA accumulated over a panel's 8 blocks and gSt over the
// region's panels in registers.  Nothing about the LDS traffic is clever, and nothing needs to be: every MFMA operand is
// ONE fp32 register per lane, fetched by a ds_read_b32 from row-major images with odd leading dimensions (conflict-free
// along rows and along columns), two reads per 64-cycle MFMA.
//   contraction order: the 32x32x2 instruction takes two contraction indices per step, one from lanes 0-31 and one from
//   lanes 32-63; step j pairs index j with index j + half, so that each lane walks a contiguous run (its half of the range).
// LDS (K = 64): S 8 x 32 x 65 + A 128 x 65 + R 2 x 128 x 33 floats = 133 KB.
// MFMAs per block and SIMD: 32 (producer) + 64 (consumer) at K = 64, 16 + 32 at K = 32.
// CHAIN: gA summed in place along chains of workgroups on one XCD, exactly as in k_grad_f16_v8<.., CHAIN> (same region map,
// rotation of the panels, arrival words with the writer's XCC_ID, fault -> the host falls back to slabs): the 64 gA slabs
// of cfg3 (268 MB written here, read back by the update kernel) become 2.
// Outputs as every K1: one gA slab per column region (per chain with CHAIN), one gSt slab per row region ([r4]: the consumers split the block's 128
// rows: halves at K = 64 -- two k tiles x two halves --, quarters at K = 32), a loss partial per workgroup.
// ------------------------------------------------------------------------------------------------
template <int K> struct F32pcCfg {
    static constexpr int LDK = K + 1;                 // leading dimension of the S / A images
    static constexpr int S_BLOCK = 32 * LDK;          // floats per 32-column block
    static constexpr int A_IMG = 128 * LDK;
    static constexpr int LDR = 33;
    static constexpr int R_IMG = 128 * LDR;
    static constexpr int OFF_A = 8 * S_BLOCK, OFF_R = OFF_A + A_IMG, LDS_FLOATS = OFF_R + 2 * R_IMG;
    static constexpr int KT = K / 32;                 // 32-wide k tiles
    static constexpr int S_SPLIT = 4 / KT;            // parts the block's 128 rows are split into for gSt (one per consumer wave and k tile)
};
static_assert(F32pcCfg<64>::LDS_FLOATS * 4 <= 160 * 1024, "");

template <int K, bool HASW, bool CHAIN>
__global__ __launch_bounds__(512, 2) void k_grad_f32_pc(GradArgs a, int gridX, int gridY) {
    using C = F32pcCfg<K>;
    constexpr int LDK = C::LDK, LDR = C::LDR, NCB = 8, KH = K / 2, KT = C::KT, S_SPLIT = C::S_SPLIT, MROWS = 128 / S_SPLIT;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* Simg = fsm;
    float* Aimg = fsm + C::OFF_A;
    float* Rimg = fsm + C::OFF_R;

    if (chain_halted(a.status)) return;
    k1_gram_fold(a.fold);                    // [r6] the step rule's Gram fold of THIS iteration, in the first workgroups (pmx_common.h)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    int chainId = 0, chainPos = 0;
    {
        const int lin = blockIdx.x;
        if constexpr (CHAIN) {               // the members of a chain: consecutive column regions, 8 apart in dispatch order (one XCD)
            chain_region_map(lin, a.chainL, gridX, chainId, chainPos, rowRegion, colRegion);
        } else if (gridY % 8 == 0) {                // consecutive workgroups land on consecutive XCDs: an XCD takes a band of column regions
            const int xcd = lin & 7, idx = lin >> 3;
            rowRegion = idx % gridX;
            colRegion = xcd * (gridY >> 3) + idx / gridX;
        } else {
            rowRegion = lin % gridX;
            colRegion = lin / gridX;
        }
    }
    const int row0 = rowRegion * a.RP * 128;
    const int col0 = colRegion * NCB * 32;
    int nrp = (M - row0 + 127) / 128;
    if (nrp > a.RP) nrp = a.RP;
    if (nrp < 0) nrp = 0;
    const int T = nrp * NCB;                 // blocks of this region; slots = T + 2
    const bool producer = w < 4;
    const int j = w & 3;
    float lossAcc = 0.f;
    // CHAIN: panel (t - chainPos) mod RP in the t-th place: the members of a chain reach a panel one after the other
    auto panel_at = [&](int t) {
        if constexpr (CHAIN) { const int p = t - chainPos; return p < 0 ? p + nrp : p; }
        else return t;
    };

    if (T <= 0) {                            // region outside the matrix: its gSt slab parts and loss partial are zero
        if (!producer && a.doS) {
            const int part = KT == 2 ? (j >> 1) : j, kt = KT == 2 ? (j & 1) : 0;
            float* dst = a.slabS + (int64_t)rowRegion * N * K;      // (one slab per row region: the parts are merged in LDS, below)
            for (int c = part * (NCB / S_SPLIT); c < (part + 1) * (NCB / S_SPLIT); ++c)
                for (int i = 0; i < 16; ++i) {
                    const int gn = col0 + c * 32 + tile_row(i, lane);
                    if (gn < N) dst[(int64_t)gn * K + kt * 32 + l31] = 0.f;
                }
        }
        if (tid == 0) a.lossPart[blockIdx.x] = 0.0;
        return;
    }

    {   // ---- the S rows of the region's 256 columns, once: block c, row n, component k -> Simg[c][n][k] --------------------
        constexpr int F4 = 256 * K / 4 / 512;                 // float4 per thread
        const float4* ssrc = reinterpret_cast<const float4*>(a.St + (int64_t)col0 * K);
        float4 sr[F4];
#pragma unroll
        for (int u = 0; u < F4; ++u) sr[u] = ssrc[tid + 512 * u];
#pragma unroll
        for (int u = 0; u < F4; ++u) {
            const int q = tid + 512 * u;                      // float4 index: row n = q / (K/4) of the region, k0 = 4 (q % (K/4))
            const int n = q / (K / 4), k0 = 4 * (q % (K / 4));
            float* d = Simg + (n >> 5) * C::S_BLOCK + (n & 31) * LDK + k0;
            d[0] = sr[u].x; d[1] = sr[u].y; d[2] = sr[u].z; d[3] = sr[u].w;
        }
    }

    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    if (producer) {
        // ================================ producers: P = A S and R ================================================
        f32x16 p0, p1;
        float yE[16], yO[16];
        float wE[HASW ? 16 : 1], wO[HASW ? 16 : 1];
        float afr[KH];                       // A[row][k], k = j + KH hi: this lane's half of the contraction range
        float4 areg[KH / 4];                 // the next panel's, as loaded
        const int jw = __builtin_amdgcn_readfirstlane(j);
        const float* ybase0 = a.Y + (int64_t)(row0 + jw * 32) * a.ldY + col0;
        const unsigned ylane = (unsigned)(4 * hi) * (unsigned)a.ldY + (unsigned)l31;
        auto load_Y = [&](int b, float (&y)[16]) {     // block b, clamped past the end of the region
            int brp = b >> 3;
            if (brp >= nrp) brp = nrp - 1;
            brp = panel_at(brp);
            const float* base = ybase0 + (int64_t)brp * 128 * a.ldY + (b & 7) * 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) y[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldY + ylane]);
        };
        const float* wbase0 = HASW ? a.W + (int64_t)(row0 + jw * 32) * a.ldW + col0 : nullptr;
        const unsigned wlane = HASW ? (unsigned)(4 * hi) * (unsigned)a.ldW + (unsigned)l31 : 0u;
        auto load_W = [&](int b, float (&wv)[HASW ? 16 : 1]) {
            if constexpr (HASW) {
                int brp = b >> 3;
                if (brp >= nrp) brp = nrp - 1;
                brp = panel_at(brp);
                const float* base = wbase0 + (int64_t)brp * 128 * a.ldW + (b & 7) * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldW + wlane]);
            }
        };
        auto load_A = [&](int prow) {
            const float4* src = reinterpret_cast<const float4*>(a.A + (int64_t)(prow + j * 32 + l31) * K + hi * KH);
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) areg[q] = src[q];
        };
        auto take_A = [&]() {
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) { afr[4 * q] = areg[q].x; afr[4 * q + 1] = areg[q].y; afr[4 * q + 2] = areg[q].z; afr[4 * q + 3] = areg[q].w; }
        };
        auto publish_A = [&]() {             // the current panel's rows -> A image, for the consumers' gSt contraction
            float* d = Aimg + (j * 32 + l31) * LDK + hi * KH;
#pragma unroll
            for (int q = 0; q < KH; ++q) d[q] = afr[q];
        };
        const int s_g1 = l31 * LDK + hi * KH;          // P contraction's B operand: S[n = l31][k = step + KH hi]
        const int r_w = (j * 32 + 4 * hi) * LDR + l31; // R producer: accumulator register i -> row tile_row(i) of the wave's 32
        load_A(row0 + panel_at(0) * 128);
        load_Y(0, yE);
        load_Y(1, yO);
        load_W(0, wE);
        load_W(1, wO);
        take_A();
        if (nrp > 1) load_A(row0 + panel_at(1) * 128);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();        // S images published

        // One slot.  GEMM: block s into pc.  EPI: block s-1 from pp and its Y tile -> R[(s-1) & 1].
        auto slot = [&](int s, f32x16& pc, f32x16& pp, float (&y)[16], float (&wv)[HASW ? 16 : 1], auto gemm_c, auto epi_c) {
            constexpr bool GEMM = decltype(gemm_c)::value, EPI = decltype(epi_c)::value;
            const int cb = s & 7, rp = s >> 3;
            if (cb == 2 && rp < nrp) {       // block s-2 opened this row panel: the consumers start on it in this slot
                publish_A();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (GEMM) {
                if (cb == 0 && s > 0) {      // block s opens a row panel: its A rows (requested 8 slots ago)
                    take_A();
                    if (rp + 1 < nrp) load_A(row0 + panel_at(rp + 1) * 128);
                }
                const float* Sb = Simg + cb * C::S_BLOCK + s_g1;
#pragma unroll
                for (int i = 0; i < 16; ++i) pc[i] = 0.f;
#pragma unroll
                for (int q = 0; q < KH; ++q) pc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q], Sb[q], pc, 0, 0, 0);
                // Left alone the scheduler fetches each operand right in front of the MFMA that takes it and waits for it (it
                // minimises registers): ~100 cycles of LDS latency in front of every 64-cycle MFMA.  Imposed order: the
                // reads run 8 operands ahead; the epilogue of block s-1 (3 VALU, a store and a Y request per element) is
                // dealt out between the MFMAs.
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int q = 0; q < KH; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q + 8 < KH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if constexpr (EPI) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 48 / KH + 1, 0);
                        if (q < 16) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
            }
            if constexpr (EPI) {
                float* Rb = Rimg + ((s - 1) & 1) * C::R_IMG + r_w;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float r = pp[i] - y[i];
                    if constexpr (HASW) {
                        lossAcc += wv[i] * (r * r);
                        r *= wv[i];
                    } else {
                        lossAcc += r * r;
                    }
                    Rb[((i & 3) + 8 * (i >> 2)) * LDR] = r;
                }
                load_Y(s + 1, y);            // the set is free again: Y of the block two slots on
                load_W(s + 1, wv);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            __builtin_amdgcn_s_barrier();
        };
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
        __builtin_amdgcn_s_barrier();              // (pairs with the consumers' barrier between parking and merging their gSt parts)
    } else {
        // ================================ consumers: gA and gSt of block s-2 ======================================
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // S images published

        f32x16 accS[NCB];
        f32x16 accA[KT];
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) accS[c][i] = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accA[t][i] = 0.f;
        // gSt: this wave's part of the block's rows and its k tile
        const int part = KT == 2 ? (j >> 1) : j, kt = KT == 2 ? (j & 1) : 0;
        constexpr int MH = MROWS / 2;                                    // contraction steps of the gSt part
        const int r_gA = (j * 32 + l31) * LDR + 16 * hi;                  // gA's A operand: R[m = 32 j + l31][n = step + 16 hi]
        const int s_gA = (16 * hi) * LDK + l31;                           // gA's B operand: S[n = step + 16 hi][k = 32 tile + l31]
        const int r_gS = (part * MROWS + MH * hi) * LDR + l31;            // gSt's A operand: R[m = part rows + step + MH hi][n = l31]
        const int a_gS = (part * MROWS + MH * hi) * LDK + kt * 32 + l31;  // gSt's B operand: A[m][k = 32 kt + l31]
        const int slabIdxA = CHAIN ? colRegion / a.chainL : colRegion;
        auto gA_tile = [&](int prow) { return a.slabA + (int64_t)slabIdxA * M * K + (int64_t)(prow + j * 32 + 4 * hi) * K + l31; };
        auto flush_gA = [&](int prow) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                float* p_ = gA_tile(prow) + t * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) p_[((i & 3) + 8 * (i >> 2)) * K] = accA[t][i];
            }
        };
        // ---- CHAIN state: see k_grad_f16_v8 (wait for arrival k -> tile += previous sum, fetched by sc1 loads in four
        //      pieces during the panel's last four blocks -> plain stores -> vmcnt(0) -> arrival k + 1, one slot later) -------
        ChainLink link;                      // chain_link.h: the hand-off protocol
        if constexpr (CHAIN) link.init(a.chainFlags, chainId, nrp, j, a.status, a.wstatus, lane);
        if constexpr (CHAIN) {
            if (a.chainInject && blockIdx.x == 0 && j == 0) link.fault(3);
        }
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        };
        auto consume = [&](int b, int prow, int cb, f32x16& accSc) {     // block b: column block cb of the panel at row prow
            const float* Rb = Rimg + (b & 1) * C::R_IMG;
            const float* Sb = Simg + cb * C::S_BLOCK;
            if (a.doA & 1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float rv = Rb[r_gA + q];
#pragma unroll
                    for (int t = 0; t < KT; ++t) accA[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(rv, Sb[s_gA + q * LDK + 32 * t], accA[t], 0, 0, 0);
                }
                // operand reads four steps (4 (1 + KT) registers) ahead of the MFMAs (see the producers' note)
                __builtin_amdgcn_sched_group_barrier(0x100, 4 * (1 + KT), 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, KT, 0);
                    if (q + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1 + KT, 0);
                }
            }
            if (a.doS) {
#pragma unroll
                for (int q = 0; q < MH; ++q) accSc = __builtin_amdgcn_mfma_f32_32x32x2f32(Rb[r_gS + q * LDR], Aimg[a_gS + q * LDK], accSc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);              // eight steps ahead
#pragma unroll
                for (int q = 0; q < MH; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q + 8 < MH) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
            if (!CHAIN && (a.doA & 1) && cb + 1 == NCB) {
                flush_gA(prow);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) accA[t][i] = 0.f;
            }
        };
        sync();
        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int pnl = panel_at(rp);
            const int prow = row0 + pnl * 128;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (cb == 0) {               // block s-2 opens a row panel: the producers publish its A rows now
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    if constexpr (CHAIN) link.open(pnl, chainPos, a.chainL, 1, nrp, a.chainBase, (a.doA & 1) != 0);
                }
                float pv[KT][4];
                if constexpr (CHAIN) {
                    if (cb == 3) link.look();
                    if (cb >= 4 && link.cadd) {   // piece cb - 4 of the previous sum: accumulator registers 4 (cb - 4) ..
                        const float* pb = gA_tile(prow) + 8 * (cb - 4) * K;
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                pv[t][q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + q * K + 32 * t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
                }
                consume(s - 2, prow, cb, accS[cb]);
                if constexpr (CHAIN) {
                    if (cb == 3) link.wait();   // the predecessor finished this panel about a panel-time ago: normally no spin
                    if (cb >= 4 && link.cadd) {
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q) accA[t][4 * (cb - 4) + q] += pv[t][q];
                    }
                    if (cb + 1 == NCB && (a.doA & 1)) {
                        flush_gA(prow);
#pragma unroll
                        for (int t = 0; t < KT; ++t)
#pragma unroll
                            for (int i = 0; i < 16; ++i) accA[t][i] = 0.f;
                        link.flushed();
                    }
                }
                sync();
                ++s;
            }
        }
        if constexpr (CHAIN) link.publish();
        // [r4] ONE gSt slab per row region: the S_SPLIT row parts of a tile (held by S_SPLIT different waves) are summed here, through
        // the launch's own LDS (every image is dead by now: 128 KB of partial tiles fit), in the fixed order part 0, 1, ..: round 3
        // wrote S_SPLIT slabs per row region and left the sum to the update kernel -- 64 slabs = 32 MB to fold at cfg2
        {
            // tile (part, kt, c) of this wave -> fsm[((kt * S_SPLIT + part) * NCB + c) * 1024 + i * 64 + lane]
            float* park = fsm + ((kt * S_SPLIT + part) * NCB) * 1024 + lane;
            if (a.doS) {
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int i = 0; i < 16; ++i) park[c * 1024 + i * 64] = accS[c][i];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (a.doS) {
                constexpr int CB = NCB / S_SPLIT;    // blocks this wave finishes: part * CB .. (for its k tile)
                float* dst = a.slabS + (int64_t)rowRegion * N * K;
#pragma unroll
                for (int cc = 0; cc < CB; ++cc) {
                    const int c = part * CB + cc;
                    const int bcol = col0 + c * 32;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float v = 0.f;
#pragma unroll
                        for (int q = 0; q < S_SPLIT; ++q) v += fsm[((kt * S_SPLIT + q) * NCB + c) * 1024 + i * 64 + lane];
                        dst[(int64_t)(bcol + tile_row(i, lane)) * K + kt * 32 + l31] = v;
                    }
                }
            }
        }
    }
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if (lane == 0) fsm[w] = v;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < 4; ++i) s += (double)fsm[i];
            a.lossPart[blockIdx.x] = s;
        }
    }
}

// host side -----------------------------------------------------------------------------------------
// shapes the kernel takes (PMX_K1_F32PC=0: off, k_grad_f32 instead -- tuning A/B)
bool grad_f32pc_applies(int64_t M, int64_t N, int64_t K) {
    if ((K != 32 && K != 64) || M % 128 != 0 || N % 256 != 0) return false;
    return !(getenv("PMX_K1_F32PC") && atoi(getenv("PMX_K1_F32PC")) == 0);
}
GradPlan grad_plan_f32pc(int64_t M, int64_t N, int64_t K) {
    GradPlan p{};
    p.KP = (int)K;
    p.BN = 32;
    const int64_t panels = M / 128;
    p.gridY = (int)(N / 256);
    const int wantWG = getenv("PMX_K1_WGS") ? atoi(getenv("PMX_K1_WGS")) : 256;   // one resident workgroup per CU
    plan_row_regions(panels, p.gridY, wantWG, &p.RP, &p.gridX);
    p.nSlabA = p.gridY;
    p.nSlabS = p.gridX;                                  // ([r4] the row parts are merged inside the launch)
    p.ldsBytes = sizeof(float) * (K == 64 ? F32pcCfg<64>::LDS_FLOATS : F32pcCfg<32>::LDS_FLOATS);
    if (p.ldsBytes < (size_t)4 * 8 * 1024 * sizeof(float)) p.ldsBytes = (size_t)4 * 8 * 1024 * sizeof(float);   // the gSt merge parks 4 x 8 tiles (K = 32: more than its images)
    p.variant = -2;
    return p;
}
template <int K, bool HASW, bool CHAIN>
static hipError_t grad_launch_f32pc_t(const GradPlan& p, const GradArgs& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f32_pc<K, HASW, CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.ldsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f32_pc<K, HASW, CHAIN>), dim3(p.gridX * p.gridY), dim3(512), p.ldsBytes, stream, a, p.gridX, p.gridY);
    return hipGetLastError();
}
template <int K>
static hipError_t grad_launch_f32pc_k(const GradPlan& p, const GradArgs& a, hipStream_t stream) {
    const bool w = a.W != nullptr;
    if (a.chainL > 0) return w ? grad_launch_f32pc_t<K, true, true>(p, a, stream) : grad_launch_f32pc_t<K, false, true>(p, a, stream);
    return w ? grad_launch_f32pc_t<K, true, false>(p, a, stream) : grad_launch_f32pc_t<K, false, false>(p, a, stream);
}
hipError_t grad_launch_f32pc(const GradPlan& p, const GradArgs& a, hipStream_t stream) {
    return a.K == 64 ? grad_launch_f32pc_k<64>(p, a, stream) : grad_launch_f32pc_k<32>(p, a, stream);
}
