// K1 for K = 128 in the two-term fp16 arithmetic of k_grad_f16_v8 (included by pmx_api.hip after k_grad_bf16.hip, whose
// image layouts, swizzles, transposing reads and operand scales it shares).
//
// Same pass, same roles as k_grad_f16_v8 -- four producer waves (P = A S, R = P - Y, R split and parked as two fp16
// [n][m] images), four consumer waves two blocks behind (gA = R S^T, gSt = R^T A from those images), one barrier per
// 128 x 32 block -- re-blocked for twice the contraction length:
//   * a region is 128 columns (4 blocks): the two fp16 terms of its S rows for all 128 k are 64 KB of LDS, kept as two
//     "k halves" per block, each laid out exactly like v8's 32 x 64 image, so every read pattern of v8 (b128 rows for the
//     P contraction, transposing reads for gA) applies to a half unchanged;
//   * the producers' A operand (32 rows x 128 k x two terms = 64 VGPRs per lane) stays in registers for a panel as in
//     v8, but there is no room to prefetch the next panel beside it, and a panel is over after 4 blocks, not 8.  So A is
//     split into its two fp16 terms ONCE per launch by k_split_a_f16 (8 bytes per element of A: noise next to Y), and in
//     the last block of a panel each 16-k fragment is re-loaded with the next panel's rows right behind the three MFMAs
//     that were its last readers: the loads land during that block's epilogue and barrier, no conversion in the loop;
//   * the consumers hold gA for the panel (32 rows x 128 k per wave: 64 accumulator registers) and their two gSt tiles
//     of each of the 4 blocks (128 registers): everything in ONE pass over Y.  (A variant doing one 64-wide half of the
//     gradients per launch, the residual computed twice, was measured at 0.56 ms against 0.40 ms and removed.)
// LDS: S 64 KB + A images 64 KB (both halves, for gSt) + R 32 KB = the CU's 160 KB.
// MFMAs per block: 24 per producer wave, 48 per consumer wave (24 + 24 at K = 64): the consumers bound the slot here.
// [r4] gSt: consumer wave j owns ONE 32-wide k tile (k = 32 j ..) of every block and contracts over all 128 rows of the
// panel (eight 16-row steps) -- round 2/3 gave a wave two k tiles and half the rows, i.e. 128 accumulator registers for the
// region's four blocks and two partial gSt slabs per row region; now 64 registers and ONE slab per row region.  With gA's 64
// that is 128 accumulator registers per consumer (k_grad_f16_v8 holds 160), which is what makes room for
// <CHAIN>: gA summed in place along XCD-local chains of workgroups -- k_grad_f16_v8<.., CHAIN>'s protocol (arrival words
// carrying the writer's XCC_ID, sc1 fetch-and-add in the consumers' registers, fault -> slabs), re-timed for a panel of
// four blocks: arrival looked at in the panel's second block, the previous sum fetched in two halves of 32 registers during
// the third and fourth, published one slot into the next panel.  A member's panels are rotated by `chainStride` panels per
// place in the chain (2 where the region has the panels for it: the predecessor's sum is then a whole panel-time old when it
// is asked for).  gA: one slab per CHAIN (N / 128 / chainL of them) instead of one per column region.
// ------------------------------------------------------------------------------------------------
#ifndef PMX_CHAIN_PF128
#define PMX_CHAIN_PF128 0    // how far ahead the gA waves of <RS, CHAIN> request the pieces of a chain's previous sum (see the loop); the A/B builds set it
#endif
constexpr int W8_NCB = 4;
constexpr int W8_S_HALF = 2 * V5_S_TERM;            // [h][l] images of one k half of a 32-column block
constexpr int W8_SL_BYTES = 2 * W8_S_HALF;          // both halves: 16 KB per block
constexpr int W8_A_HALF = V5_AIMG_BYTES;            // [h][l] images of one k half of the 128-row panel: 32 KB
constexpr int W8_OFF_A = W8_NCB * W8_SL_BYTES;
constexpr int W8_NKT = 2;                           // 64-wide k halves
constexpr int W8_OFF_R = W8_OFF_A + W8_NKT * W8_A_HALF, W8_LDS_BYTES = W8_OFF_R + 2 * V5_R_BYTES;
static_assert(W8_LDS_BYTES <= 160 * 1024, "");
static_assert(W8_OFF_R % 256 == 0, "R images must start on a bank row");

struct GradK128Args {
    const float* Y;
    int64_t ldY;
    const _Float16* Ah;      // [M][128] high / low fp16 terms of 2^eA A (k_split_a_f16)
    const _Float16* Al;
    const float* St;         // [N][128]
    float* slabA;
    float* slabS;
    double* lossPart;
    const DevStatus* status;
    int M, N;
    int RP;
    int doA, doS;
    int gridX, gridY;
    const float* absmax;     // [2][V8_NPART] partial maxima of |A|, |St|
    float ymax;
    const float* W;          // <HASW>: weights of the likelihood (nmf.py:13-41), M x N, row pitch ldW; nullptr: W == 1
    int64_t ldW;
    float wmax;              // max(1, max |W|): enters the bound that scales R
    // <CHAIN> (see k_grad_f16_v8<.., CHAIN>, GradV4Args)
    int chainL;              // workgroups per chain (0: one gA slab per column region)
    int chainStride;         // panels between the rotations of neighbouring members (chainStride * chainL <= RP)
    unsigned* chainFlags;    // [chains][RP][4] arrival words, monotonic over launches
    unsigned chainBase;      // launch sequence number * 64
    DevStatus* wstatus;      // writable view of `status` (fault report)
    int chainInject;         // tests: report a fault from this launch
    float rangeRatio;        // [r4] f16_range_fault (k_grad_f16_v8.hip); 0: no check
    int hh;                  // [r5] <.., HH>: the residual from the high x high product alone, the rest as a correction slab (k_gfix.hip; no weights)
    int consPrio;            // [r6] s_setprio level of the consumer waves (k1_set_priority, k_grad_f16_v8.hip)
};

struct SplitAArgs {
    const float* X;          // [count] fp32
    int64_t count;           // multiple of 8
    const float* absmax;     // [V8_NPART] partial maxima of |X|
    _Float16* H;
    _Float16* L;
    const DevStatus* status;
};
// 2^e with max|X| 2^e in [2^13, 2^14) from the partial maxima (the same expression k_grad_f16_k128 evaluates)
__device__ __forceinline__ float w8_scale_from_partials(const float* part, float* red, int tid, int nthreads) {
    float m = 0.f;
    for (int i = tid; i < V8_NPART; i += nthreads) m = fmaxf(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    float mx = red[0];
    for (int i = 1; i < nthreads / 64; ++i) mx = fmaxf(mx, red[i]);
    __syncthreads();
    int q = 0;
    (void)frexpf(mx, &q);
    return ldexpf(1.f, mx > 0.f ? 14 - q : 0);
}
__global__ __launch_bounds__(256) void k_split_a_f16(SplitAArgs a) {
    __shared__ float red[4];
    if (chain_halted(a.status)) return;
    const float sc = w8_scale_from_partials(a.absmax, red, threadIdx.x, 256);
    const int64_t n8 = a.count >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 x0 = reinterpret_cast<const float4*>(a.X)[2 * i], x1 = reinterpret_cast<const float4*>(a.X)[2 * i + 1];
        f16x4 h0, l0, h1, l1;
        v8_split2(x0, sc, h0, l0);
        v8_split2(x1, sc, h1, l1);
        f16x8 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = h0[q]; h[4 + q] = h1[q]; l[q] = l0[q]; l[4 + q] = l1[q]; }
        reinterpret_cast<f16x8*>(a.H)[i] = h;
        reinterpret_cast<f16x8*>(a.L)[i] = l;
    }
}
void launch_split_a_f16(const SplitAArgs& a, hipStream_t s) {
    int64_t blocks = (a.count / 8 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_split_a_f16, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// HASW: weighted likelihood -- D = W (A S - Y), loss 1/2 sum W (Y - A S)^2: the producers fetch the W tile next to the Y tile
// (they have the registers: the consumers bound this kernel) and scale R in the epilogue, as in k_grad_f16_v8<.., HASW>.
// HH [r5]: k_grad_f16_v8<.., HH>'s residual (P0 = a0 s0: 8 MFMAs per block instead of 24; what it leaves out arrives as a correction slab, k_gfix.hip):
// the mode-f16x2r instance of this kernel -- there is no LDS here for third terms, and none are needed.
// RS [r5]: the consumers' roles split by contraction, as in k_grad_f16_v8<.., RS> (why: there): waves 0, 1 contract gA for 64 rows each (eight accumulator tiles;
// the block's S fragments are read twice per slot instead of four times), waves 2, 3 contract gSt for TWO k tiles each over all 128 rows of the panel (the R^T
// fragments are read twice per slot instead of four times; the high terms of the panel's A fragments stay in registers for the panel's four slots).  48 MFMAs
// per wave and slot as before; LDS reads per slot 112 KB instead of 208 KB.  Gradient passes that want both gradients, <HH> only.
template <bool HASW, bool CHAIN, bool HH = false, bool RS = false>
__global__ __launch_bounds__(V5_THREADS, 2) void k_grad_f16_k128(GradK128Args a) {
    static_assert(!(HH && HASW), "HH: unweighted contexts");
    static_assert(!RS || HH, "RS: an instance of the <HH> gradient pass");
    constexpr int K = 128, ROWB = 128, NCB = W8_NCB, NKT = W8_NKT;
    constexpr int OFF_R = W8_OFF_R;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lq = lane >> 4;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    int chainId = 0, chainPos = 0;           // CHAIN: which chain, and this workgroup's place in its rotation
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if constexpr (CHAIN) {               // the members of a chain are multiples of 8 apart in dispatch order: ONE XCD (checked at run time)
            chain_region_map(lin, a.chainL, gx, chainId, chainPos, rowRegion, colRegion);
        } else if (gy % 8 == 0) {            // consecutive workgroups land on consecutive XCDs: an XCD takes a band of column regions
            const int xcd = lin & 7, idx = lin >> 3;
            rowRegion = idx % gx;
            colRegion = xcd * (gy >> 3) + idx / gx;
        } else {
            rowRegion = lin % gx;
            colRegion = lin / gx;
        }
    }
    const int row0 = rowRegion * a.RP * V5_BM;
    const int col0 = colRegion * NCB * V5_BN;
    int nrp = (M - row0 + V5_BM - 1) / V5_BM;
    if (nrp > a.RP) nrp = a.RP;
    if (nrp < 0) nrp = 0;
    const int T = nrp * NCB;                 // blocks of this region; slots = T + 2
    const bool producer = w < 4;
    const int j = w & 3;
    float lossAcc = 0.f;
    // CHAIN: the t-th panel this workgroup visits is panel (t - chainStride * chainPos) mod RP (every region has all RP panels
    // in this mode): the members of a chain are on different panels at any time, and member c reaches a panel chainStride
    // panel-times after member c - 1 did
    const int rot = CHAIN ? a.chainStride * chainPos : 0;
    auto panel_at = [&](int t) {
        if constexpr (CHAIN) { const int p = t - rot; return p < 0 ? p + nrp : p; }
        else return t;
    };

    if (T <= 0) {                            // region outside the matrix: its gSt slab parts and loss partial are zero
        if (!producer) {
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
            for (int c = 0; c < NCB; ++c)
                for (int i = 0; i < 16; ++i) {
                    const int gn = col0 + c * V5_BN + tile_row(i, lane);
                    if (gn < N && a.doS) dst[(int64_t)gn * K + j * 32 + l31] = 0.f;
                }
        }
        if (tid == 0) a.lossPart[blockIdx.x] = 0.0;
        return;
    }

    // ---- power-of-two operand scales (see k_grad_f16_v8); uniform --------------------------------------------------
    float scS, scR, unP, unA, unS;
    {
        float* red = reinterpret_cast<float*>(smem);
        float m0 = 0.f, m1 = 0.f;
        for (int i = tid; i < V8_NPART; i += V5_THREADS) { m0 = fmaxf(m0, a.absmax[i]); m1 = fmaxf(m1, a.absmax[V8_NPART + i]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
        if (lane == 0) { red[w] = m0; red[8 + w] = m1; }
        __syncthreads();
        float mA = red[0], mS = red[8];
        for (int i = 1; i < 8; ++i) { mA = fmaxf(mA, red[i]); mS = fmaxf(mS, red[8 + i]); }
        __syncthreads();                     // red aliases the S images
        if (f16_range_fault((float)K * mA * mS, a.ymax, a.rangeRatio, a.doA, a.doS, a.wstatus, tid)) return;
        int qA = 0, qS = 0, qR = 0;
        (void)frexpf(mA, &qA);
        (void)frexpf(mS, &qS);
        (void)frexpf((a.ymax + (float)K * mA * mS) * (HASW ? a.wmax : 1.f), &qR);
        const int eA = mA > 0.f ? 14 - qA : 0, eS = mS > 0.f ? 14 - qS : 0, eR = 14 - qR;
        scS = ldexpf(1.f, eS); scR = ldexpf(1.f, eR);
        unP = ldexpf(1.f, -(eA + eS)); unA = ldexpf(1.f, -(eR + eS)); unS = ldexpf(1.f, -(eR + eA));
    }
    {   // ---- both fp16 terms of the region's 128 S rows, once: block c, k half (k >> 6) -> its v8-style image ----------
        const float4* ssrc = reinterpret_cast<const float4*>(a.St + (int64_t)col0 * K) + tid;
        float4 sr[NCB][2];
#pragma unroll
        for (int c = 0; c < NCB; ++c) { sr[c][0] = ssrc[c * 1024]; sr[c][1] = ssrc[c * 1024 + 512]; }
        const int c4 = tid & 31, half = c4 >> 4, cc = c4 & 15;
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = (tid >> 5) + 16 * u;
                f16x4 t0, t1;
                v8_split2(sr[c][u], scS, t0, t1);
                unsigned char* d = smem + c * W8_SL_BYTES + half * W8_S_HALF + row * ROWB + ((((cc >> 1) ^ v3_swz(row)) & 7) << 4) + 8 * (cc & 1);
                *reinterpret_cast<f16x4*>(d) = t0;
                *reinterpret_cast<f16x4*>(d + V5_S_TERM) = t1;
            }
    }

    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    if (producer) {
        k1_set_priority(-a.consPrio);      // (PMX_K1_PRIO < 0: the producers instead -- A/B only)
        // ================================ producers: P = A S and R ================================================
        f32x16 p0, p1;
        float yE[16], yO[16];
        float wv1[HASW ? 16 : 1];            // weights of ONE block, requested a slot ahead of their use (a slot is ~3 us here)
        f16x8 afr[8][2];
        const int jw = __builtin_amdgcn_readfirstlane(j);
        const float* ybase0 = a.Y + (int64_t)(row0 + jw * 32) * a.ldY + col0;
        // saddr-form requests: the row's address is scalar (block base + row pitch), the lane's part ONE 32-bit byte offset
        // held for the whole launch -- no 64-bit address pairs in VGPRs (16 of them were live around every batch of requests)
        const unsigned ylane = ((unsigned)(4 * hi) * (unsigned)a.ldY + (unsigned)l31) * 4u;
        auto load_Y = [&](int b, float (&y)[16]) {     // block b, clamped past the end of the region
            int brp = b >> 2;
            if (brp >= nrp) brp = nrp - 1;
            brp = panel_at(brp);
            const float* base = ybase0 + (int64_t)brp * V5_BM * a.ldY + (b & 3) * V5_BN;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned vo = ylane;
                asm volatile("" : "+v"(vo));     // opaque copy: the compiler keeps (scalar row address) + (32-bit VGPR offset) = the saddr form
                const unsigned char* rb = reinterpret_cast<const unsigned char*>(base + (int64_t)((i & 3) + 8 * (i >> 2)) * a.ldY);
                y[i] = __builtin_nontemporal_load(reinterpret_cast<const float*>(rb + vo));
            }
        };
        const float* wbase0 = HASW ? a.W + (int64_t)(row0 + jw * 32) * a.ldW + col0 : nullptr;
        const unsigned wlane = HASW ? ((unsigned)(4 * hi) * (unsigned)a.ldW + (unsigned)l31) * 4u : 0u;
        auto load_W = [&](int b, float (&wv)[HASW ? 16 : 1]) {
            if constexpr (HASW) {
                int brp = b >> 2;
                if (brp >= nrp) brp = nrp - 1;
                brp = panel_at(brp);
                const float* base = wbase0 + (int64_t)brp * V5_BM * a.ldW + (b & 3) * V5_BN;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    unsigned vo = wlane;
                    asm volatile("" : "+v"(vo));
                    const unsigned char* rb = reinterpret_cast<const unsigned char*>(base + (int64_t)((i & 3) + 8 * (i >> 2)) * a.ldW);
                    wv[i] = __builtin_nontemporal_load(reinterpret_cast<const float*>(rb + vo));
                }
            }
        };
        // fragment ks of this lane: k = 16 ks + 8 hi .. + 7 of row (panel row 32 j + l31)
        const int64_t afrag0 = (int64_t)(row0 + j * 32 + l31) * K + hi * 8;
        auto load_afr = [&](int rp, int ks) {      // (past the region's last panel: that panel again, so that every panel ends alike)
            if (rp >= nrp) rp = nrp - 1;
            rp = panel_at(rp);
            const int64_t o = afrag0 + (int64_t)rp * V5_BM * K + ks * 16;
            afr[ks][0] = *reinterpret_cast<const f16x8*>(a.Ah + o);
            afr[ks][1] = *reinterpret_cast<const f16x8*>(a.Al + o);
        };
        auto publish_A = [&]() {             // the current panel's terms -> A images, for the consumers' gSt contraction
            const int pa = (j * 32 + l31) * ROWB + ((hi ^ v3_swz(j * 32 + l31)) << 4);   // chunk 2 k4 + hi: ^ (k4 << 5)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                unsigned char* d = smem + W8_OFF_A + (ks >> 2) * W8_A_HALF + (pa ^ ((ks & 3) << 5));
                *reinterpret_cast<f16x8*>(d) = afr[ks][0];
                *reinterpret_cast<f16x8*>(d + V5_A_TERM) = afr[ks][1];
            }
        };
        const int s_g1 = l31 * ROWB + ((hi ^ v3_swz(l31)) << 4);                 // P contraction's B operand: row l31, chunk 2 k4 + hi
        const int r_w = l31 * 256 + (((4 * j) ^ v4_swz(l31)) << 4) + 8 * hi;      // R producer, ^ (g << 4)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) load_afr(0, ks);
        load_W(0, wv1);                      // (stands in for the weights of "block -1": R = 0 there; the loop head sees the
                                             // same requests in flight from here as from the end of a panel)
        load_Y(0, yE);
        // slot 0 runs the same code as every other slot (no peeled copy: the loop head then sees the same requests in flight
        // from both sides and the compiler's wait counts stay exact): its epilogue works on a zero "block -1" -- R = 0 into
        // an image nobody reads before block 1 rewrites it, nothing added to the loss -- and requests Y(1)
#pragma unroll
        for (int i = 0; i < 16; ++i) { p1[i] = 0.f; yO[i] = 0.f; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();        // S images published

        // One slot (block s = (rp, cb)).  GEMM: block s into pc.  EPI: block s-1 from pp and its Y tile -> R[(s-1) & 1].
        // RELOAD: the panel's last block: each A fragment is re-loaded with the next panel's rows behind its MFMAs (one
        // code path for every panel: two variants would meet at the loop head with different numbers of requests in
        // flight, and the compiler's merged wait counts then drain the Y tiles requested a slot ago).
        auto slot = [&](int s, int rp, auto cb_c, f32x16& pc, f32x16& pp, float (&y)[16], float (&wv)[HASW ? 16 : 1], auto gemm_c, auto epi_c, auto reload_c) {
            constexpr int cb = decltype(cb_c)::value;
            constexpr bool GEMM = decltype(gemm_c)::value, EPI = decltype(epi_c)::value, RELOAD = decltype(reload_c)::value;
            if constexpr (cb == 2 && GEMM) {         // block s-2 opened this row panel: the consumers start on it in this slot
                publish_A();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (GEMM) {
                const unsigned char* Slb = smem + cb * W8_SL_BYTES;
#pragma unroll
                for (int i = 0; i < 16; ++i) pc[i] = 0.f;
                f16x8 sh[8], sl[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int so = (ks >> 2) * W8_S_HALF + (s_g1 ^ ((ks & 3) << 5));
                    sh[ks] = *reinterpret_cast<const f16x8*>(Slb + so);
                    if constexpr (!HH) sl[ks] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if constexpr (!HH) {
                        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][1], sh[ks], pc, 0, 0, 0);
                        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sl[ks], pc, 0, 0, 0);
                    }
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sh[ks], pc, 0, 0, 0);
                    if constexpr (RELOAD) {
                        // pinned in program order (only VALU / SALU / LDS instructions may move across): behind the MFMAs
                        // that read the old fragment (else both live: spills), and AHEAD of this slot's Y requests -- the
                        // next slot opens by waiting for these fragments (L2 hits), and vmcnt counts in order: it must
                        // not wait for the Y tiles (HBM latency) requested a moment ago
                        __builtin_amdgcn_sched_barrier(0x86);
                        load_afr(rp + 1, ks);
                        __builtin_amdgcn_sched_barrier(0x86);
                    }
                }
                if constexpr (!RELOAD) {
                    // The 24 MFMAs are one dependent chain; left alone the scheduler (the kernel as a whole sits at the
                    // register limit) reads one S fragment pair, waits for it, issues its three MFMAs, reads the next pair
                    // into the same registers ... : eight exposed LDS latencies per slot.  Order imposed here: the reads
                    // run two fragment pairs (24 VGPRs) ahead of the MFMAs that use them.
                    constexpr int NR = HH ? 1 : 2, NM = HH ? 1 : 3;       // LDS reads / MFMAs per k step
                    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
                        if (ks + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
                        if constexpr (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);   // the epilogue of block s-1 in the MFMAs' shadow
                    }
                }
            }
            if constexpr (EPI) {
                unsigned char* Rb = smem + OFF_R + ((s - 1) & 1) * V5_R_BYTES;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 h, l;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r = pp[4 * g + q] * unP - y[4 * g + q];
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
                    }
                    const int o = r_w ^ (g << 4);
                    *reinterpret_cast<f16x4*>(Rb + o) = h;
                    *reinterpret_cast<f16x4*>(Rb + V5_R_TERM + o) = l;
                }
                load_W(s, wv);               // weights of block s: the next slot's epilogue.  AHEAD of the Y requests: vmcnt counts
                                             // in order, and waiting for these must not mean waiting for a Y tile with a slot to spare
                load_Y(s + 1, y);            // the set is free again: Y of the block two slots on
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            __builtin_amdgcn_s_barrier();
            // The four slots of a panel are ONE basic block (compile-time cb).  Without a fence the scheduler pulls the head
            // of the next slot's epilogue up here -- and with it a wait for the Y tile requested one slot ago.
            __builtin_amdgcn_sched_barrier(0);
        };
        using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
        using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
        // even blocks: accumulator p0, Y set yE; odd blocks: p1, yO.  Slot s requests Y(s + 1) into the set block s - 1 has left.
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int s = rp * NCB;
            slot(s, rp, c0{}, p0, p1, yO, wv1, yes{}, yes{}, no{});
            slot(s + 1, rp, c1{}, p1, p0, yE, wv1, yes{}, yes{}, no{});
            slot(s + 2, rp, c2{}, p0, p1, yO, wv1, yes{}, yes{}, no{});
            slot(s + 3, rp, c3{}, p1, p0, yE, wv1, yes{}, yes{}, yes{});
        }
        slot(T, nrp, c0{}, p0, p1, yO, wv1, no{}, yes{}, no{});
        slot(T + 1, nrp, c1{}, p1, p0, yE, wv1, no{}, no{}, no{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (RS) {
        // ================================ consumers, roles split by contraction (see RS in the header) ==============
        k1_set_priority(a.consPrio);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // S images published
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        };
        auto tr_src = [&](int row, int k0) {
            const int kk = k0 + 16 * (lq & 1) + 4 * (li & 3);
            return row * ROWB + ((((kk >> 3) ^ v3_swz(row)) & 7) << 4) + 8 * ((kk >> 2) & 1);
        };
        using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
        using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
        if (j < 2) {
            // ---- gA: rows 64 j .. 64 j + 63 of the panel (row tiles rt), all four k tiles (h * 64 + t * 32) ---------------------
            f32x16 accA[2][NKT][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int h = 0; h < NKT; ++h)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { accA[rt][h][0][i] = 0.f; accA[rt][h][1][i] = 0.f; }
            int r_t[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int m = (2 * j + rt) * 32 + 16 * (lq & 1) + 4 * (li & 3);
                const int n0 = 8 * hi + (li >> 2), n1 = n0 + 4;
                r_t[rt][0] = n0 * 256 + ((((m >> 3) ^ v4_swz(n0)) & 15) << 4) + 8 * ((m >> 2) & 1);
                r_t[rt][1] = n1 * 256 + ((((m >> 3) ^ v4_swz(n1)) & 15) << 4) + 8 * ((m >> 2) & 1);
            }
            const int s_t0 = tr_src(8 * hi + (li >> 2), 0), s_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);
            const int slabIdxA = CHAIN ? colRegion / a.chainL : colRegion;
            auto gA_tile = [&](int prow, int rt) { return a.slabA + (int64_t)slabIdxA * M * K + (int64_t)(prow + (2 * j + rt) * 32 + 4 * hi) * K + l31; };
            auto flush_gA = [&](int prow) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int h = 0; h < NKT; ++h) {
                        float* p0_ = gA_tile(prow, rt) + h * 64;
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            float* ph_ = p0_ + half * 16 * K;
                            asm volatile("" : "+v"(ph_));
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int i = half * 8 + q;
                                const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                                ph_[ro] = accA[rt][h][0][i] * unA;
                                ph_[ro + 32] = accA[rt][h][1][i] * unA;
                            }
                        }
                    }
            };
            auto consumeA_ks = [&](int b, auto cb_c, auto ks_c) {      // one of the block's two steps of sixteen columns
                constexpr int cb = decltype(cb_c)::value, ks = decltype(ks_c)::value;
                const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
                const unsigned char* Slb = smem + cb * W8_SL_BYTES;
                {
                    f16x8 r0[2], r1[2];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        r0[rt] = v8_tr_pair(Rb, r_t[rt][0] + ks * 4096, r_t[rt][1] + ks * 4096);
                        r1[rt] = v8_tr_pair(Rb + V5_R_TERM, r_t[rt][0] + ks * 4096, r_t[rt][1] + ks * 4096);
                    }
                    const int so0 = s_t0 + ks * 16 * ROWB, so1 = s_t1 + ks * 16 * ROWB;
#pragma unroll
                    for (int h = 0; h < NKT; ++h) {
                        const unsigned char* Sh = Slb + h * W8_S_HALF;
                        const f16x8 s00 = v8_tr_pair(Sh, so0, so1);
                        const f16x8 s01 = v8_tr_pair(Sh + V5_S_TERM, so0, so1);
                        const f16x8 s10 = v8_tr_pair(Sh, so0 ^ 64, so1 ^ 64);
                        const f16x8 s11 = v8_tr_pair(Sh + V5_S_TERM, so0 ^ 64, so1 ^ 64);
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            accA[rt][h][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1[rt], s00, accA[rt][h][0], 0, 0, 0);
                            accA[rt][h][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1[rt], s10, accA[rt][h][1], 0, 0, 0);
                            accA[rt][h][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0[rt], s01, accA[rt][h][0], 0, 0, 0);
                            accA[rt][h][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0[rt], s11, accA[rt][h][1], 0, 0, 0);
                            accA[rt][h][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0[rt], s00, accA[rt][h][0], 0, 0, 0);
                            accA[rt][h][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0[rt], s10, accA[rt][h][1], 0, 0, 0);
                        }
                    }
                }
            };
            auto consumeA = [&](int b, auto cb_c) { consumeA_ks(b, cb_c, c0{}); consumeA_ks(b, cb_c, c1{}); };
            const float invUnA = scR * scS;
            ChainLink link;
            if constexpr (CHAIN) link.init(a.chainFlags, chainId, nrp, j, a.status, a.wstatus, lane);
            // piece (rt, h) of the previous sum of this wave's rows (32 registers): requested in front of half a block's MFMAs, added behind them
            auto chain_fetch = [&](int prow, int rt, int h, float (&pv)[32]) {
                const float* pb = gA_tile(prow, rt) + h * 64;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float* q = pb + ((i & 3) + 8 * (i >> 2)) * K;
                    pv[i] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    pv[16 + i] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(q + 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
            };
            auto chain_add = [&](int rt, int h, const float (&pv)[32]) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    accA[rt][h][0][i] += pv[i] * invUnA;
                    accA[rt][h][1][i] += pv[16 + i] * invUnA;
                }
            };
            if constexpr (CHAIN) {
                if (a.chainInject && blockIdx.x == 0 && j == 0) link.fault(3);
            }
            sync();
            sync();
            int s = 2;
#pragma nounroll
            for (int rp = 0; rp < nrp; ++rp) {
                const int pnl = panel_at(rp);
                const int prow = row0 + pnl * V5_BM;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                if constexpr (CHAIN) link.open(pnl, chainPos, a.chainL, a.chainStride, nrp, a.chainBase, (a.doA & 1) != 0);
                consumeA(s - 2, c0{}); sync(); ++s;
                if constexpr (CHAIN) link.look();
                consumeA(s - 2, c1{});
                if constexpr (CHAIN) link.wait();
                // [r6] PMX_CHAIN_PF128: how far ahead of its add a piece of the previous sum is requested (the adds stand where they stood: same sums, same order).
                //   0: in front of the half block of MFMAs it is added behind (rounds 4-5: 24 MFMAs = ~0.4 us per round trip)
                //   1: one buffer, every piece requested as soon as the buffer is free -- the first behind the arrival check, the third behind the second's add
                //   2: two buffers (+ 32 registers): two pieces in flight, a whole slot per round trip
                if constexpr (CHAIN && PMX_CHAIN_PF128 == 2) {
                    float pa[32], pb[32];
                    if (link.cadd) { chain_fetch(prow, 0, 0, pa); chain_fetch(prow, 0, 1, pb); }
                    sync(); ++s;
                    consumeA_ks(s - 2, c2{}, c0{});
                    if (link.cadd) { chain_add(0, 0, pa); chain_fetch(prow, 1, 0, pa); }
                    consumeA_ks(s - 2, c2{}, c1{});
                    if (link.cadd) { chain_add(0, 1, pb); chain_fetch(prow, 1, 1, pb); }
                    sync(); ++s;
                    consumeA_ks(s - 2, c3{}, c0{});
                    if (link.cadd) chain_add(1, 0, pa);
                    consumeA_ks(s - 2, c3{}, c1{});
                    if (link.cadd) chain_add(1, 1, pb);
                } else if constexpr (CHAIN && PMX_CHAIN_PF128 == 1) {
                    float pv[32];
                    if (link.cadd) chain_fetch(prow, 0, 0, pv);
                    sync(); ++s;
                    consumeA_ks(s - 2, c2{}, c0{});
                    if (link.cadd) { chain_add(0, 0, pv); chain_fetch(prow, 0, 1, pv); }
                    consumeA_ks(s - 2, c2{}, c1{});
                    if (link.cadd) { chain_add(0, 1, pv); chain_fetch(prow, 1, 0, pv); }
                    sync(); ++s;
                    consumeA_ks(s - 2, c3{}, c0{});
                    if (link.cadd) { chain_add(1, 0, pv); chain_fetch(prow, 1, 1, pv); }
                    consumeA_ks(s - 2, c3{}, c1{});
                    if (link.cadd) chain_add(1, 1, pv);
                } else if constexpr (CHAIN) {
                    sync(); ++s;
                    float pv[32];
                    if (link.cadd) chain_fetch(prow, 0, 0, pv);
                    consumeA_ks(s - 2, c2{}, c0{});
                    if (link.cadd) { chain_add(0, 0, pv); chain_fetch(prow, 0, 1, pv); }
                    consumeA_ks(s - 2, c2{}, c1{});
                    if (link.cadd) chain_add(0, 1, pv);
                    sync(); ++s;
                    if (link.cadd) chain_fetch(prow, 1, 0, pv);
                    consumeA_ks(s - 2, c3{}, c0{});
                    if (link.cadd) { chain_add(1, 0, pv); chain_fetch(prow, 1, 1, pv); }
                    consumeA_ks(s - 2, c3{}, c1{});
                    if (link.cadd) chain_add(1, 1, pv);
                } else {
                    sync(); ++s;
                    consumeA(s - 2, c2{}); sync(); ++s;
                    consumeA(s - 2, c3{});
                }
                flush_gA(prow);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int h = 0; h < NKT; ++h)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { accA[rt][h][0][i] = 0.f; accA[rt][h][1][i] = 0.f; }
                if constexpr (CHAIN) link.flushed();
                sync(); ++s;
            }
            if constexpr (CHAIN) link.publish();
        } else {
            // ---- gSt: k tiles 2 (j - 2) and 2 (j - 2) + 1 of every block of the region, all 128 rows of a panel ----------------------
            const int kh = j - 2;                // the 64-wide half of the A images these two tiles live in
            f32x16 accS[NCB][2];
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) { accS[c][0][i] = 0.f; accS[c][1][i] = 0.f; }
            const int r_g3 = l31 * 256 + ((hi ^ v4_swz(l31)) << 4);
            int a_t[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { a_t[t][0] = tr_src(8 * hi + (li >> 2), t * 32); a_t[t][1] = tr_src(8 * hi + 4 + (li >> 2), t * 32); }
            const unsigned char* Ab = smem + W8_OFF_A + kh * W8_A_HALF;
            f16x8 af[8][2];                      // the panel's A fragments, HIGH terms (the low terms are read per slot: registers)
            auto consumeS = [&](int b, auto cb_c) {
                constexpr int cb = decltype(cb_c)::value;
                const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f16x8 a1 = v8_tr_pair(Ab + V5_A_TERM, a_t[t][0] + ks * 16 * ROWB, a_t[t][1] + ks * 16 * ROWB);
                        accS[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, af[ks][t], accS[cb][t], 0, 0, 0);
                        accS[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, accS[cb][t], 0, 0, 0);
                        accS[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, af[ks][t], accS[cb][t], 0, 0, 0);
                    }
                }
            };
            sync();
            sync();
            int s = 2;
#pragma nounroll
            for (int rp = 0; rp < nrp; ++rp) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();          // the producers have published the panel's A terms
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t) af[ks][t] = v8_tr_pair(Ab, a_t[t][0] + ks * 16 * ROWB, a_t[t][1] + ks * 16 * ROWB);
                consumeS(s - 2, c0{}); sync(); ++s;
                consumeS(s - 2, c1{}); sync(); ++s;
                consumeS(s - 2, c2{}); sync(); ++s;
                consumeS(s - 2, c3{}); sync(); ++s;
            }
            {
                float* dst = a.slabS + (int64_t)rowRegion * N * K;
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const int bcol = col0 + c * V5_BN;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int kk = (2 * kh + t) * 32 + l31;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int gn = bcol + tile_row(i, lane);
                            dst[(int64_t)gn * K + kk] = accS[c][t][i] * unS;
                        }
                    }
                }
            }
        }
    } else {
        // ================================ consumers: gA and gSt of block s-2 ======================================
        k1_set_priority(a.consPrio);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // S images published

        f32x16 accS[NCB];                    // gSt: k tile 32 j .. of each of the region's four blocks, all 128 rows of every panel
        f32x16 accA[NKT][2];                 // gA: rows 32 j .. of the panel, k tiles h * 64 + t * 32
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) accS[c][i] = 0.f;
#pragma unroll
        for (int h = 0; h < NKT; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) { accA[h][0][i] = 0.f; accA[h][1][i] = 0.f; }
        const int kt = j & 1, kh = j >> 1;   // gSt's k tile: half kh of the A images, 32-wide tile kt within it
        int r_t0, r_t1;                      // gA's A operand (R, transposing read)
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
        const int s_t0 = tr_src(8 * hi + (li >> 2), 0), s_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);   // gA's B operand; k tile 1: ^ 64
        const int r_g3 = l31 * 256 + ((hi ^ v4_swz(l31)) << 4);                                // gSt's A operand (R^T rows n), rows 16 ks ..: ^ (ks << 5)
        const int a_t0 = tr_src(8 * hi + (li >> 2), kt * 32), a_t1 = tr_src(8 * hi + 4 + (li >> 2), kt * 32);   // gSt's B operand, rows 16 ks ..: + ks * 16 * ROWB
        // gA slab this workgroup contributes to: its own (one per column region), or its chain's (accumulated in place)
        const int slabIdxA = CHAIN ? colRegion / a.chainL : colRegion;
        auto gA_tile = [&](int prow) { return a.slabA + (int64_t)slabIdxA * M * K + (int64_t)(prow + j * 32 + 4 * hi) * K + l31; };
        auto flush_gA = [&](int prow) {
#pragma unroll
            for (int h = 0; h < NKT; ++h) {
                float* p0_ = gA_tile(prow) + h * 64;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float* ph_ = p0_ + half * 16 * K;
                    asm volatile("" : "+v"(ph_));          // keep it ONE pointer: the offsets below fold into the store's immediate
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = half * 8 + q;        // tile_row(i) = (i & 3) + 8 * (i >> 2) + 4 * hi
                        const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                        ph_[ro] = accA[h][0][i] * unA;
                        ph_[ro + 32] = accA[h][1][i] * unA;
                    }
                }
            }
        };
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        };
        auto consumeA = [&](int b, auto cb_c) {    // gA of block b: column block cb of the current panel
            constexpr int cb = decltype(cb_c)::value;
            const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
            const unsigned char* Slb = smem + cb * W8_SL_BYTES;
            if (a.doA & 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f16x8 r0 = v8_tr_pair(Rb, r_t0 + ks * 4096, r_t1 + ks * 4096);
                    const f16x8 r1 = v8_tr_pair(Rb + V5_R_TERM, r_t0 + ks * 4096, r_t1 + ks * 4096);
                    const int so0 = s_t0 + ks * 16 * ROWB, so1 = s_t1 + ks * 16 * ROWB;
#pragma unroll
                    for (int h = 0; h < NKT; ++h) {
                        const unsigned char* Sh = Slb + h * W8_S_HALF;
                        const f16x8 s00 = v8_tr_pair(Sh, so0, so1);
                        const f16x8 s01 = v8_tr_pair(Sh + V5_S_TERM, so0, so1);
                        const f16x8 s10 = v8_tr_pair(Sh, so0 ^ 64, so1 ^ 64);
                        const f16x8 s11 = v8_tr_pair(Sh + V5_S_TERM, so0 ^ 64, so1 ^ 64);
                        accA[h][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s00, accA[h][0], 0, 0, 0);
                        accA[h][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s10, accA[h][1], 0, 0, 0);
                        accA[h][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s01, accA[h][0], 0, 0, 0);
                        accA[h][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s11, accA[h][1], 0, 0, 0);
                        accA[h][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s00, accA[h][0], 0, 0, 0);
                        accA[h][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s10, accA[h][1], 0, 0, 0);
                    }
                }
            }
        };
        auto consumeS = [&](int b, auto cb_c) {    // gSt of block b
            constexpr int cb = decltype(cb_c)::value;
            const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
            const unsigned char* Ab = smem + W8_OFF_A + kh * W8_A_HALF;
            if (a.doS) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {       // rows 16 ks .. of the panel
                    const int ro = r_g3 ^ (ks << 5);
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
                    const f16x8 a0 = v8_tr_pair(Ab, ao0, ao1);
                    const f16x8 a1 = v8_tr_pair(Ab + V5_A_TERM, ao0, ao1);
                    accS[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, accS[cb], 0, 0, 0);
                    accS[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, accS[cb], 0, 0, 0);
                    accS[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, accS[cb], 0, 0, 0);
                }
            }
        };
        auto consume = [&](int b, auto cb_c) { consumeA(b, cb_c); consumeS(b, cb_c); };
        // ---- CHAIN: gA summed in place through the XCD's L2 (protocol and fault handling: chain_link.h) --------------------
        const float invUnA = scR * scS;              // 2^(eR+eS): previous sums enter the accumulators in their scale
        ChainLink link;
        if constexpr (CHAIN) link.init(a.chainFlags, chainId, nrp, j, a.status, a.wstatus, lane);
        // k half h of the previous sum of this wave's tile (32 registers): requested in front of a block's MFMAs, added behind them
        auto chain_fetch = [&](int prow, int h, float (&pv)[32]) {
            const float* pb = gA_tile(prow) + h * 64;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* q = pb + ((i & 3) + 8 * (i >> 2)) * K;
                pv[i] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                pv[16 + i] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(q + 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        };
        auto chain_add = [&](int h, const float (&pv)[32]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                accA[h][0][i] += pv[i] * invUnA;
                accA[h][1][i] += pv[16 + i] * invUnA;
            }
        };
        if constexpr (CHAIN) {
            if (a.chainInject && blockIdx.x == 0 && j == 0) link.fault(3);
        }
        using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
        using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
        sync();
        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int pnl = panel_at(rp);
            const int prow = row0 + pnl * V5_BM;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();          // block s-2 opens a row panel: the producers have published its A terms
            if constexpr (CHAIN) link.open(pnl, chainPos, a.chainL, a.chainStride, nrp, a.chainBase, (a.doA & 1) != 0);
            consume(s - 2, c0{}); sync(); ++s;
            if constexpr (CHAIN) link.look();
            consume(s - 2, c1{});
            if constexpr (CHAIN) link.wait();
            sync(); ++s;
            // The panel's last block: gA first, so that its flush (64 stores per lane) has the block's gSt contraction and the
            // barrier to land behind -- the arrival word is published at the top of the next panel behind a vmcnt(0) that then
            // costs nothing.  CHAIN: the previous sum arrives in two halves of 32 registers, each requested in front of MFMAs.
            if constexpr (CHAIN) {
                float pv[32];
                if (link.cadd) chain_fetch(prow, 0, pv);
                consume(s - 2, c2{});
                if (link.cadd) chain_add(0, pv);
                sync(); ++s;
                if (link.cadd) chain_fetch(prow, 1, pv);
                consumeA(s - 2, c3{});
                if (link.cadd) chain_add(1, pv);
            } else {
                consume(s - 2, c2{}); sync(); ++s;
                consumeA(s - 2, c3{});
            }
            if (a.doA & 1) {
                flush_gA(prow);
#pragma unroll
                for (int h = 0; h < NKT; ++h)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { accA[h][0][i] = 0.f; accA[h][1][i] = 0.f; }
                if constexpr (CHAIN) link.flushed();
            }
            __builtin_amdgcn_sched_barrier(0);     // (the stores stay in front of the gSt contraction)
            consumeS(s - 2, c3{});
            sync(); ++s;
        }
        if constexpr (CHAIN) link.publish();
        if (a.doS) {
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
            const int kk = j * 32 + l31;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int bcol = col0 + c * V5_BN;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gn = bcol + tile_row(i, lane);
                    dst[(int64_t)gn * K + kk] = accS[c][i] * unS;
                }
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
            for (int i = 0; i < 4; ++i) s += (double)red[i];
            a.lossPart[blockIdx.x] = s;
        }
    }
}

// host side -----------------------------------------------------------------------------------------
// shapes the kernel takes (PMX_K1_K128=0: off, the exact-fp32 kernel instead -- tuning A/B)
bool grad_k128_applies(int64_t M, int64_t N, int64_t K) {
    if (K != 128 || M % V5_BM != 0 || N % (W8_NCB * V5_BN) != 0) return false;
    return !(getenv("PMX_K1_K128") && atoi(getenv("PMX_K1_K128")) == 0);
}
GradPlan grad_plan_k128(int64_t M, int64_t N) {
    GradPlan p{};
    p.KP = 128;
    p.BN = V5_BN;
    const int64_t panels = M / V5_BM;
    p.gridY = (int)(N / (W8_NCB * V5_BN));
    const int wantWG = getenv("PMX_K1_WGS") ? atoi(getenv("PMX_K1_WGS")) : 256;   // one resident workgroup per CU
    plan_row_regions(panels, p.gridY, wantWG, &p.RP, &p.gridX);
    p.nSlabA = p.gridY;
    p.nSlabS = p.gridX;
    p.ldsBytes = W8_LDS_BYTES;
    return p;
}
// panels between the rotations of neighbouring chain members (PMX_K128_STRIDE: A/B)
int grad_k128_chain_stride(const GradPlan& p, int chainL) {
    if (chainL <= 0) return 1;
    const int want = getenv("PMX_K128_STRIDE") ? atoi(getenv("PMX_K128_STRIDE")) : 1;   // measured (profiles/r04_a_k128_chain_ab.txt): 1 beats 2 -- the predecessor's sum is still in L2
    int sg = p.RP / chainL;
    if (sg > want) sg = want;
    return sg < 1 ? 1 : sg;
}
template <bool HASW, bool CHAIN, bool HH = false, bool RS = false>
static hipError_t grad_launch_k128_t(const GradK128Args& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f16_k128<HASW, CHAIN, HH, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f16_k128<HASW, CHAIN, HH, RS>), dim3(a.gridX * a.gridY), dim3(V5_THREADS), W8_LDS_BYTES, stream, a);
    return hipGetLastError();
}
hipError_t grad_launch_k128(const GradK128Args& a, hipStream_t stream) {
    if (a.hh && a.W == nullptr && (a.doA & 1) && a.doS && !(getenv("PMX_K1_ROLE_SPLIT") && atoi(getenv("PMX_K1_ROLE_SPLIT")) == 0))    // <RS>: both gradients wanted (PMX_K1_ROLE_SPLIT=0: A/B)
        return a.chainL > 0 ? grad_launch_k128_t<false, true, true, true>(a, stream) : grad_launch_k128_t<false, false, true, true>(a, stream);
    if (a.hh && a.W == nullptr && ((a.doA & 1) || a.doS))      // (the loss-only pass has nowhere to put a correction: two terms)
        return a.chainL > 0 && (a.doA & 1) ? grad_launch_k128_t<false, true, true>(a, stream) : grad_launch_k128_t<false, false, true>(a, stream);
    if (a.chainL > 0 && (a.doA & 1))
        return a.W != nullptr ? grad_launch_k128_t<true, true>(a, stream) : grad_launch_k128_t<false, true>(a, stream);
    return a.W != nullptr ? grad_launch_k128_t<true, false>(a, stream) : grad_launch_k128_t<false, false>(a, stream);
}
