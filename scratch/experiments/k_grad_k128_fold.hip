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
// gA: one slab per column region (N / 128 of them), gSt: two per row region, as v8 without chaining.
//
// <FOLD>: the gA slabs are as many bytes as Y, and the update kernel that sums them afterwards spends ~100 us of a 0.54 ms
// rank-iteration (cfg4's 8192-row share) streaming them back in.  In this instance the PRODUCER waves (they have registers
// and issue slots to spare: the consumers bound the slot) sum the tiles while K1 is still running, a few panels behind their
// writers: workgroup (row region r, column region c) folds row c of every 128-row panel of r (for 128 column regions; rows
// c, c + G, .. for G = 64, 32), producer wave w the tiles of column regions [w G/4, (w+1) G/4), 8 loads of 8 bytes per lane
// and slot, requested behind the slot's Y requests and added one slot later, at the very end of the slot -- nothing on the critical path waits for them.  The consumers'
// flush stores are agent-scope (write-through: the reader sits on another XCD), and one slot after a flush each consumer
// wave stores the launch number into its arrival word; a producer wave polls the 32 words it depends on with ONE load per
// slot (same latency) and only requests tiles of panels it has seen complete; requests of a wave that has nothing
// ready go to a page of zeros, so the instruction stream -- and with it every vmcnt the compiler counts -- is the same in
// every slot.  What is left after the last panel (two or three panels) is folded in a polling loop with a 20 ms bound:
// workgroups that are not co-resident make it expire -> DevStatus::k1_fault, the chain of kernels stops before anything is
// updated and the host repeats the iteration without FOLD (pmx_api.hip, as for the chained K1 of K = 64).  The update kernels
// then sum FOUR slabs (one per producer wave) instead of N / 128: deterministic, fixed order.
// ------------------------------------------------------------------------------------------------
#ifndef PMX_FOLD_ORDER
#define PMX_FOLD_ORDER 1
#endif
#ifndef PMX_FOLD_ABL
#define PMX_FOLD_ABL 0
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
    float* foldA;            // <FOLD>: [4][M][128] gA summed over column regions [w G/4, (w+1) G/4) by producer wave w, then gridX gridY 4 dummy rows
    unsigned* foldFlags;     // <FOLD>: [gridX][RP][4][gridY]: launch number of the last flush by (row region, panel, consumer wave, column region)
    const float* foldZero;   // <FOLD>: 512 bytes of zeros
    unsigned foldSeq;        // <FOLD>: this launch's number (> 0)
    int foldInject;          // tests: report a fault
    DevStatus* wstatus;
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
template <bool HASW, bool FOLD>
__global__ __launch_bounds__(V5_THREADS, 2) void k_grad_f16_k128(GradK128Args a) {
    constexpr int K = 128, ROWB = 128, NCB = W8_NCB, NKT = W8_NKT;
    constexpr int OFF_R = W8_OFF_R;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lq = lane >> 4;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if (gy % 8 == 0) {                   // consecutive workgroups land on consecutive XCDs: an XCD takes a band of column regions
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

    if (T <= 0) {                            // region outside the matrix: its gSt slab parts and loss partial are zero
        if (!producer) {
            const int mh = j >> 1, kt = j & 1;
            float* dst = a.slabS + (int64_t)(rowRegion * 2 + mh) * N * K;
            for (int h = 0; h < NKT; ++h)
                for (int c = 0; c < NCB; ++c)
                    for (int i = 0; i < 16; ++i) {
                        const int gn = col0 + c * V5_BN + tile_row(i, lane);
                        if (gn < N && a.doS) dst[(int64_t)gn * K + h * 64 + kt * 32 + l31] = 0.f;
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
        // ================================ producers: P = A S and R ================================================
        f32x16 p0, p1;
        float yE[16], yO[16];
        float wv1[HASW ? 16 : 1];            // weights of ONE block, requested a slot ahead of their use (a slot is ~3 us here)
        f16x8 afr[8][2];
        const int jw = __builtin_amdgcn_readfirstlane(j);
        const float* ybase0 = a.Y + (int64_t)(row0 + jw * 32) * a.ldY + col0;
        // saddr-form requests: the row's address is scalar (block base + row pitch), the lane's part ONE 32-bit byte offset
        // held for the whole launch -- no 64-bit address pairs in VGPRs (16 of them were live around every batch of requests)
        unsigned ylane = ((unsigned)(4 * hi) * (unsigned)a.ldY + (unsigned)l31) * 4u;
        auto load_Y = [&](int b, float (&y)[16]) {     // block b, clamped past the end of the region
            int brp = b >> 2;
            if (brp >= nrp) brp = nrp - 1;
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
        const unsigned wlane = HASW ? (unsigned)(4 * hi) * (unsigned)a.ldW + (unsigned)l31 : 0u;
        auto load_W = [&](int b, float (&wv)[HASW ? 16 : 1]) {
            if constexpr (HASW) {
                int brp = b >> 2;
                if (brp >= nrp) brp = nrp - 1;
                const float* base = wbase0 + (int64_t)brp * V5_BM * a.ldW + (b & 3) * V5_BN;
#pragma unroll
                for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(&base[(int64_t)((i & 3) + 8 * (i >> 2)) * a.ldW + wlane]);
            }
        };
        // fragment ks of this lane: k = 16 ks + 8 hi .. + 7 of row (panel row 32 j + l31)
        const int64_t afrag0 = (int64_t)(row0 + j * 32 + l31) * K + hi * 8;
        auto load_afr = [&](int rp, int ks) {      // (past the region's last panel: that panel again, so that every panel ends alike)
            if (rp >= nrp) rp = nrp - 1;
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
        // ---- <FOLD>: this wave's share of the in-kernel gA sum (see the head of the file) ----------------------------
        // One "quarter" per slot: 8 tiles' worth of ONE row (512 B each: 8 B per lane).  G = gridY column regions, G / 4 of
        // them per wave, 128 / G rows per workgroup and panel, G / 32 quarters per row, four quarters per panel.
        const int G = a.gridY, qs = G == 128 ? 2 : (G == 64 ? 1 : 0), qpr = 1 << qs;
        int f_panel = 0, f_q = 0;                // next request: quarter f_q of panel f_panel
        int f_ready = 0;                         // panels whose tiles (this wave's column regions) are known to have landed
        float f_acc0 = 0.f, f_acc1 = 0.f;        // running sum of the row in progress
        unsigned long long fset[2][8];
        unsigned fpoll[2] = {0u, 0u};
        int fpoll_panel[2] = {-1, -1};
        float f_keep[2] = {1.f, 1.f};
        int64_t f_dst[2] = {0, 0};
        const int64_t f_dummy = FOLD ? ((int64_t)4 * M + (int64_t)blockIdx.x * 4 + jw) * K : 0;
        const int64_t f_zoff = FOLD ? a.foldZero - a.slabA : 0;
        const unsigned f_voff = (unsigned)lane * 8u;   // byte offset of this lane's two floats in a 512-byte row
        // arrival words this wave depends on: lane l & 31 <-> (row c + G ri, column region jw G/4 + cr): written by consumer wave (row >> 5)
        int f_flagoff = 0;
        if constexpr (FOLD) {
            const int l = lane & 31, per = G >> 2, ri = l / per, cr = l % per;
            f_flagoff = ((colRegion + G * ri) >> 5) * G + jw * per + cr;
        }
        bool f_dead = false;
        auto fold_fault = [&](int code) {
            if (lane == 0) {
                a.wstatus->k1_fault = code;
                a.wstatus->reason = HALT_ERROR;
                __threadfence();
                a.wstatus->halt = 1;
            }
            f_dead = true;
        };
        auto fold_consume = [&](auto ic) {       // the set requested a slot ago
            constexpr int i = decltype(ic)::value;
            float s0 = f_acc0, s1 = f_acc1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += __builtin_bit_cast(float, (unsigned)(fset[i][u] & 0xffffffffull));
                s1 += __builtin_bit_cast(float, (unsigned)(fset[i][u] >> 32));
            }
            float2 out = {s0, s1};
            {
                unsigned vo = f_voff;
                asm volatile("" : "+v"(vo));
                *reinterpret_cast<float2*>(reinterpret_cast<unsigned char*>(a.foldA + f_dst[i]) + vo) = out;   // a finished row, or the dummy row
            }
            const float keep = f_keep[i];
            f_acc0 = s0 * keep; f_acc1 = s1 * keep;
            const unsigned long long okm = __builtin_amdgcn_ballot_w64((int)(fpoll[i] - a.foldSeq) >= 0);
            f_ready += (int)(okm == ~0ull) & (int)(fpoll_panel[i] == f_ready);
        };
        auto fold_request = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            // selections by mask arithmetic: a conditional expression here becomes a branch, and a branch ends the basic block
            // the slot's hand-laid schedule (and the compiler's exact wait counts) live in
            const int64_t vmask = (PMX_FOLD_ABL & 2) ? 0 : -(int64_t)(f_panel < f_ready);             // all ones: this quarter's tiles have landed
            const int ri = f_q >> qs, qq = f_q & (qpr - 1);
            const int fp = min(f_panel, nrp - 1);
            const int row = row0 + fp * V5_BM + colRegion + G * ri;
            const int64_t realoff = ((int64_t)(jw * (G >> 2) + qq * 8) * M + row) * K;
            const int64_t off = f_zoff + ((realoff - f_zoff) & vmask);       // the page of zeros, as an offset from slabA
            const int64_t stride = ((int64_t)M * K) & vmask;
            const float* p = a.slabA + off;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                unsigned vo = f_voff;
                asm volatile("" : "+v"(vo));
                const unsigned char* pu = reinterpret_cast<const unsigned char*>(p + u * stride) + vo;
                fset[i][u] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int64_t emask = vmask & -(int64_t)(qq == qpr - 1);         // this quarter completes a row
            f_keep[i] = __builtin_bit_cast(float, (unsigned)(~emask) & 0x3f800000u);   // 0.f after a finished row, else 1.f
            f_dst[i] = f_dummy + ((((int64_t)jw * M + row) * K - f_dummy) & emask);
            const int adv = (int)(vmask & 1);
            f_panel += (f_q + adv) >> 2;
            f_q = (f_q + adv) & 3;
            const int pp = min(f_ready, nrp - 1);
            fpoll[i] = __hip_atomic_load(a.foldFlags + ((int64_t)(rowRegion * a.RP + pp) * 4) * G + f_flagoff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fpoll_panel[i] = pp;
        };
        if constexpr (FOLD) {
            // the loop head must see the same requests in flight from the prologue as from the end of a panel (the compiler
            // merges the two and its wait counts stay exact only then): a first request -- nothing is ready, it goes to the
            // page of zeros -- in front of Y(0), as every slot has one in front of its Y requests
            f_dst[0] = f_dst[1] = f_dummy;
            if (a.foldInject && blockIdx.x == 0 && jw == 0) fold_fault(3);
            fold_request(std::integral_constant<int, 0>{});
        }
        load_Y(0, yE);
        // slot 0 runs the same code as every other slot (no peeled copy: the loop head then sees the same requests in flight
        // from both sides and the compiler's wait counts stay exact): its epilogue works on a zero "block -1" -- R = 0 into
        // an image nobody reads before block 1 rewrites it, nothing added to the loss -- and requests Y(1)
#pragma unroll
        for (int i = 0; i < 16; ++i) { p1[i] = 0.f; yO[i] = 0.f; if constexpr (HASW) wv1[i] = 0.f; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();        // S images published

        // One slot (block s = (rp, cb)).  GEMM: block s into pc.  EPI: block s-1 from pp and its Y tile -> R[(s-1) & 1].
        // RELOAD: the panel's last block: each A fragment is re-loaded with the next panel's rows behind its MFMAs (one
        // code path for every panel: two variants would meet at the loop head with different numbers of requests in
        // flight, and the compiler's merged wait counts then drain the Y tiles requested a slot ago).
        auto slot = [&](int s, int rp, auto cb_c, f32x16& pc, f32x16& pp, float (&y)[16], float (&wv)[HASW ? 16 : 1], auto gemm_c, auto epi_c, auto reload_c) {
            constexpr int cb = decltype(cb_c)::value;
            constexpr bool GEMM = decltype(gemm_c)::value, EPI = decltype(epi_c)::value, RELOAD = decltype(reload_c)::value;
            if constexpr (FOLD && EPI) {
                // the epilogue's arithmetic is pure: nothing but this keeps instruction selection from starting it at the end of
                // the PREVIOUS slot (behind the sums' code) -- and with it a wait for the Y tile requested a slot ago
                asm volatile("" : "+v"(pp));
            }
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
                auto read_S = [&](int ks) {
                    const int so = (ks >> 2) * W8_S_HALF + (s_g1 ^ ((ks & 3) << 5));
                    sh[ks] = *reinterpret_cast<const f16x8*>(Slb + so);
                    sl[ks] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                };
                if constexpr (RELOAD && FOLD) {
                    read_S(0); read_S(1);
                } else {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) read_S(ks);
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][1], sh[ks], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sl[ks], pc, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sh[ks], pc, 0, 0, 0);
                    if constexpr (RELOAD) {
                        // pinned in program order (only VALU / SALU / LDS instructions may move across): behind the MFMAs
                        // that read the old fragment (else both live: spills), and AHEAD of this slot's Y requests -- the
                        // next slot opens by waiting for these fragments (L2 hits), and vmcnt counts in order: it must
                        // not wait for the Y tiles (HBM latency) requested a moment ago.
                        // <FOLD> (a kernel at the register limit): the S fragments are pinned as well, two pairs ahead of
                        // their MFMAs -- left free, all sixteen are read up front (64 VGPRs) and the loop spills
                        __builtin_amdgcn_sched_barrier(FOLD ? 0x06 : 0x86);
                        load_afr(rp + 1, ks);
                        if constexpr (FOLD) { if (ks + 2 < 8) read_S(ks + 2); }
                        __builtin_amdgcn_sched_barrier(FOLD ? 0x06 : 0x86);
                    }
                }
                if constexpr (!RELOAD) {
                    // The 24 MFMAs are one dependent chain; left alone the scheduler (the kernel as a whole sits at the
                    // register limit) reads one S fragment pair, waits for it, issues its three MFMAs, reads the next pair
                    // into the same registers ... : eight exposed LDS latencies per slot.  Order imposed here: the reads
                    // run two fragment pairs (24 VGPRs) ahead of the MFMAs that use them.
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        if (ks + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
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
#if PMX_FOLD_ORDER == 1
                if constexpr (FOLD) {        // AHEAD of the Y requests: vmcnt counts in order, and the sums' requests must not be
                    __builtin_amdgcn_sched_barrier(0);      // younger than a Y tile that has two slots to arrive
                    fold_consume(std::integral_constant<int, 0>{});     // ONE set in flight: requested a slot ago
                    fold_request(std::integral_constant<int, 0>{});
                }
#endif
                load_Y(s + 1, y);            // the set is free again: Y of the block two slots on
                load_W(s, wv);               // weights of block s: the next slot's epilogue
            }
            if constexpr (FOLD && (!EPI || PMX_FOLD_ORDER == 0)) {
                __builtin_amdgcn_sched_barrier(0);
                fold_consume(std::integral_constant<int, 0>{});
                fold_request(std::integral_constant<int, 0>{});
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
        if constexpr (FOLD) {
            // the two sets still in flight, then whatever is left (the last two or three panels), polling: every workgroup of
            // the row region must get there -- 20 ms bound, like the chained K1's hand-off
            fold_consume(std::integral_constant<int, 0>{});
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (f_panel < nrp && !f_dead && !(PMX_FOLD_ABL & 2)) {
                const int before = f_panel * 4 + f_q;
                fold_request(std::integral_constant<int, 0>{});
                fold_consume(std::integral_constant<int, 0>{});
                if (f_panel * 4 + f_q == before && f_ready <= f_panel) {
                    __builtin_amdgcn_s_sleep(16);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) fold_fault(1);
                }
            }
        }
    } else {
        // ================================ consumers: gA and gSt of block s-2 ======================================
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // S images published

        f32x16 accS[NCB][NKT];
        f32x16 accA[NKT][2];
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int h = 0; h < NKT; ++h)
#pragma unroll
                for (int i = 0; i < 16; ++i) accS[c][h][i] = 0.f;
#pragma unroll
        for (int h = 0; h < NKT; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) { accA[h][0][i] = 0.f; accA[h][1][i] = 0.f; }
        const int kt = j & 1, mh = j >> 1;   // gSt tile (per k half); gA: rows 32 j .., both 32-wide k tiles of each half
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
        const int r_g3 = l31 * 256 + (((8 * mh + hi) ^ v4_swz(l31)) << 4);                     // gSt's A operand, ^ (ks << 5)
        const int a_t0 = tr_src(64 * mh + 8 * hi + (li >> 2), kt * 32), a_t1 = tr_src(64 * mh + 8 * hi + 4 + (li >> 2), kt * 32);   // gSt's B operand
        auto flush_gA = [&](int prow) {
#pragma unroll
            for (int h = 0; h < NKT; ++h) {
                float* p0_ = a.slabA + (int64_t)colRegion * M * K + (int64_t)(prow + j * 32 + 4 * hi) * K + h * 64 + l31;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float* ph_ = p0_ + half * 16 * K;
                    asm volatile("" : "+v"(ph_));          // keep it ONE pointer: the offsets below fold into the store's immediate
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = half * 8 + q;        // tile_row(i) = (i & 3) + 8 * (i >> 2) + 4 * hi
                        const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                        if constexpr (FOLD && !(PMX_FOLD_ABL & 1)) {              // agent scope: written through to where a reader on another XCD finds it
                            __hip_atomic_store(reinterpret_cast<unsigned*>(ph_ + ro), __builtin_bit_cast(unsigned, accA[h][0][i] * unA), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(reinterpret_cast<unsigned*>(ph_ + ro + 32), __builtin_bit_cast(unsigned, accA[h][1][i] * unA), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            ph_[ro] = accA[h][0][i] * unA;
                            ph_[ro + 32] = accA[h][1][i] * unA;
                        }
                    }
                }
            }
        };
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        };
        unsigned* pendFlag = nullptr;           // <FOLD>: arrival word to write once this wave's flush stores have been acknowledged
        auto fold_publish = [&]() {
            if constexpr (FOLD) {
                if (pendFlag != nullptr) {
                    if (!(PMX_FOLD_ABL & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(pendFlag, a.foldSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::: "memory");
                    pendFlag = nullptr;
                }
            }
        };
        auto consume = [&](int b, int prow, auto cb_c) {     // block b: column block cb of the panel at row prow
            constexpr int cb = decltype(cb_c)::value;
            const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
            const unsigned char* Slb = smem + cb * W8_SL_BYTES;
            const unsigned char* Ab = smem + W8_OFF_A;
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
            if (a.doS) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
#pragma unroll
                    for (int h = 0; h < NKT; ++h) {
                        const unsigned char* Ahh = Ab + h * W8_A_HALF;
                        const f16x8 a0 = v8_tr_pair(Ahh, ao0, ao1);
                        const f16x8 a1 = v8_tr_pair(Ahh + V5_A_TERM, ao0, ao1);
                        accS[cb][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, accS[cb][h], 0, 0, 0);
                        accS[cb][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, accS[cb][h], 0, 0, 0);
                        accS[cb][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, accS[cb][h], 0, 0, 0);
                    }
                }
            }
            if constexpr (cb == 0) fold_publish();          // the previous panel's flush is one slot old: its stores have landed
            if constexpr (cb + 1 == NCB) {
                if (a.doA & 1) {
                    flush_gA(prow);
                    if constexpr (FOLD) pendFlag = a.foldFlags + ((int64_t)(rowRegion * a.RP + (prow - row0) / V5_BM) * 4 + j) * a.gridY + colRegion;
#pragma unroll
                    for (int h = 0; h < NKT; ++h)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { accA[h][0][i] = 0.f; accA[h][1][i] = 0.f; }
                }
            }
        };
        using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
        using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
        sync();
        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int prow = row0 + rp * V5_BM;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();          // block s-2 opens a row panel: the producers have published its A terms
            consume(s - 2, prow, c0{}); sync(); ++s;
            consume(s - 2, prow, c1{}); sync(); ++s;
            consume(s - 2, prow, c2{}); sync(); ++s;
            consume(s - 2, prow, c3{}); sync(); ++s;
        }
        fold_publish();
        if (a.doS) {
            float* dst = a.slabS + (int64_t)(rowRegion * 2 + mh) * N * K;
#pragma unroll
            for (int h = 0; h < NKT; ++h) {
                const int kk = h * 64 + kt * 32 + l31;
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const int bcol = col0 + c * V5_BN;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int gn = bcol + tile_row(i, lane);
                        dst[(int64_t)gn * K + kk] = accS[c][h][i] * unS;
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
    int64_t wantX = (wantWG + p.gridY - 1) / p.gridY;
    if (wantX < 1) wantX = 1;
    if (wantX > panels) wantX = panels;
    p.RP = (int)((panels + wantX - 1) / wantX);
    p.gridX = (int)((panels + p.RP - 1) / p.RP);
    p.nSlabA = p.gridY;
    p.nSlabS = p.gridX * 2;
    p.ldsBytes = W8_LDS_BYTES;
    return p;
}
template <bool HASW, bool FOLD>
static hipError_t grad_launch_k128_t(const GradK128Args& a, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f16_k128<HASW, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f16_k128<HASW, FOLD>), dim3(a.gridX * a.gridY), dim3(V5_THREADS), W8_LDS_BYTES, stream, a);
    return hipGetLastError();
}
// the in-kernel gA sum (<FOLD>): every workgroup has all its panels, 32 / 64 / 128 column regions, one resident workgroup per CU
bool grad_k128_fold_applies(const GradPlan& p, int64_t M, int ncu) {
    if (getenv("PMX_K1_K128_FOLD") && atoi(getenv("PMX_K1_K128_FOLD")) == 0) return false;
    if (p.gridY != 32 && p.gridY != 64 && p.gridY != 128) return false;
    if (M % ((int64_t)p.RP * V5_BM) != 0 || (int64_t)p.gridX * p.RP * V5_BM != M) return false;
    return ncu > 0 && p.gridX * p.gridY <= ncu;
}
hipError_t grad_launch_k128(const GradK128Args& a, hipStream_t stream) {
    // (no weighted <FOLD> instance: the weights' 16 registers on top of the sums' 20 make the producers' loop spill, and a
    // spill reload drains the Y requests in flight; pmx_api.hip leaves the sum to the update kernels for a weighted context)
    if (a.W != nullptr) return grad_launch_k128_t<true, false>(a, stream);
    const bool fold = a.foldA != nullptr && (a.doA & 1);
    return fold ? grad_launch_k128_t<false, true>(a, stream) : grad_launch_k128_t<false, false>(a, stream);
}
