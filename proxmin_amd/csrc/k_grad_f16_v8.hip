// K1 for K = 64 in the two-term fp16 arithmetic: k_grad_f16_v8 (mode f16x2: bench.py's default kernel) with its helpers, the
// factor-maxima kernels its operand scales come from and its launch wrappers.  Included by k_grad_bf16.hip, whose frame (region
// map, image layouts, swizzles, transposing reads, GradV4Args, the chain protocol of k_grad_bf16_v7<.., CHAIN>) it shares.
// ------------------------------------------------------------------------------------------------
// two-term fp16 split (k_grad_f16_v8): helpers and the factor maxima its operand scales come from
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#ifndef PMX_CHAIN_PF
#define PMX_CHAIN_PF 4       // pieces of a chain's previous sum in flight (k_grad_f16_v8<.., RS>'s gA waves): all four requested behind the arrival check (profiles/r06_q_chain_prefetch_ab.txt); 0 = rounds 2-5
#endif
constexpr int V8_NPART = 256;                // partial maxima per factor

__device__ __forceinline__ void v8_split2(const float4& x, float sc, f16x4& h, f16x4& l) {
    const float v[4] = {x.x * sc, x.y * sc, x.z * sc, x.w * sc};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 t = (_Float16)v[i];
        h[i] = t;
        l[i] = (_Float16)(v[i] - (float)t);
    }
}
// three fp16 terms of x sc (R3: the residual's operands): h = fp16(v), l = fp16(v - h), m = fp16(v - h - l) -- v - h and (v - h) - l are
// exact in fp32, so the three terms carry v to 2^-33 of its magnitude (down to fp16's subnormal floor, 2^-38 of the scaled maximum)
__device__ __forceinline__ void v8_split3(float x, float sc, _Float16& h, _Float16& l, _Float16& m) {
    const float v = x * sc;
    h = (_Float16)v;
    const float e1 = v - (float)h;
    l = (_Float16)e1;
    m = (_Float16)(e1 - (float)l);
}
// (h, l) of the pair (r0 sc, r1 sc), packed: h = fp16(x), l = fp16(x - h) with x - h formed by ONE mixed-precision fma that reads
// its fp16 operand directly (v_fma_mix*: fp32 product r sc -- exact, sc is a power of two -- minus h, rounded once: the same
// value as fp16(x - float(h)), whose difference is exact in fp32).  Left to itself hipcc converts h back to fp32, subtracts
// with a packed fp32 fma and converts again: five instructions per pair instead of three.
__device__ __forceinline__ void v8_split_pair(float r0, float r1, float sc, unsigned& h, unsigned& l) {
    unsigned hh;                             // h = fp16(r sc): the same instruction with a zero addend (one rounding)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(r0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(r1), "v"(sc));
    h = hh;
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(r0), "v"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(r1), "v"(sc), "v"(h));
    l = lo;
}
__device__ __forceinline__ f16x8 v8_tr_pair(const unsigned char* base, int off0, int off1) {
    return __builtin_bit_cast(f16x8, v3_tr_pair(base, off0, off1));   // the transposing read moves 16-bit payloads
}

// [r4] The residual R = P - Y is scaled for fp16 by a BOUND, max|Y| + K max|A| max|S|: one power of two for the whole launch.  An
// entry of R keeps both of its fp16 terms down to 2^-28 of that bound and is flushed to zero below 2^-38 of it.  Where the model
// term is far above the data -- factors that ran away (RAdam's unrectified first steps with nmf.step_adaprox reach 1e6 within two
// iterations of a unit-scale problem: bound / max|Y| ~ 2^40), components scaled apart by many orders of magnitude -- the entries
// with P = 0 (rows of A / columns of S the prox has set to zero: R = -Y there, and the whole gradient of such a row comes from
// them) fall below that.  fp32 carries an exponent per entry and does not care.  The kernels therefore refuse to run when
// K max|A| max|S| > ratio * max|Y| (ratio 2^16: R = -Y entries down to 2^-8 max|Y| keep both terms): every workgroup returns
// before anything is written, workgroup 0 reports DevStatus::k1_fault = 4 and halts the chain of kernels; the host continues
// the SAME iteration with the exact-fp32 kernel of the frame for the rest of the context's life (pmx_api.hip: k1_leave_f16).
// Gradient passes only: the loss-only instance sums fp32 residuals before the split.  Uniform over the grid (same inputs).
__device__ __forceinline__ bool f16_range_fault(float boundP, float ymax, float ratio, int doA, int doS, DevStatus* wst, int tid) {
    if (!(ratio > 0.f) || wst == nullptr || !(doA | doS) || !(ymax > 0.f) || !(boundP > ratio * ymax)) return false;
    if (blockIdx.x == 0 && tid == 0) {
        wst->k1_fault = 4;
        wst->reason = HALT_ERROR;
        __threadfence();
        wst->halt = 1;
    }
    return true;
}

// [r6] static priority for one role of the workgroup (MI355X_MICROARCH.md "two waves per SIMD", item 4): each SIMD hosts one producer and one consumer wave;
// the consumers (24 MFMAs + their LDS operand reads per slot) are the pole, the producers wait at the barrier.  Arbitration is by priority, then age --
// and the consumers are the YOUNGER half.  One s_setprio for the launch, no per-slot flips.  Level from the launch arguments (PMX_K1_PRIO: A/B).
__device__ __forceinline__ void k1_set_priority(int level) {
    if (level == 1) __builtin_amdgcn_s_setprio(1);
    else if (level == 2) __builtin_amdgcn_s_setprio(2);
    else if (level >= 3) __builtin_amdgcn_s_setprio(3);
}

// absmax[f * V8_NPART + b] = max |X_f| over workgroup b's share (f = 0: A, M x 64; f = 1: St, N x 64)
struct AbsmaxArgs {
    const float* X[2];
    int64_t count[2];        // elements (multiples of 4)
    float* out;              // [2][V8_NPART]
    const DevStatus* status;
};
__global__ __launch_bounds__(256) void k_absmax(AbsmaxArgs a) {
    __shared__ float red[4];
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    const float4* x = reinterpret_cast<const float4*>(a.X[f]);
    const int64_t n4 = a.count[f] >> 2;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)V8_NPART * 256) {
        const float4 v = x[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) a.out[f * V8_NPART + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// max |Y| of an M x N matrix with row pitch ld: out[b] = partial of workgroup b (V8_NPART of them)
__global__ __launch_bounds__(256) void k_absmax_pitched(const float* Y, int64_t ld, int64_t M, int64_t N, float* out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t r = blockIdx.x; r < M; r += V8_NPART) {
        const float* row = Y + r * ld;
        for (int64_t c = threadIdx.x; c < N; c += 256) m = fmaxf(m, fabsf(row[c]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
void launch_absmax_pitched(const float* Y, int64_t ld, int64_t M, int64_t N, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_absmax_pitched, dim3(V8_NPART), dim3(256), 0, s, Y, ld, M, N, out);
}
void launch_absmax(const AbsmaxArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_absmax, dim3(V8_NPART, 2), dim3(256), 0, s, a); }

// ------------------------------------------------------------------------------------------------
// k_grad_f16_v8 (K = 64, M % 128 == 0, N % 256 == 0): v7's kernel with TWO-term fp16 splits instead of three-term bf16.
//
// fp16 carries 11 significant bits, so x ~ (h + l) with h = fp16(x), l = fp16(x - h) is good to 2^-22 -- as long as l
// stays out of fp16's subnormal range, which ordinary NMF factors (|x| < 2^-3) would not.  Every operand is therefore
// scaled by a power of two (exact) before the split:
//   A by 2^eA, S by 2^eS with max|A| 2^eA, max|S| 2^eS in [2^13, 2^14)   (maxima of the CURRENT factors: k_absmax)
//   R by 2^eR with (max|Y| + K max|A| max|S|) max(1, max|W|) 2^eR < 2^14   (a bound: W R can never overflow fp16)
// Entries far below the maximum lose RELATIVE precision in their low term, but their absolute error stays below
// 2^-25 of the scaled range (2^-39 of the maximum), which is what matters inside a dot product.  Products:
//   A S      ah sh + ah sl + al sh      (the dropped al sl is 2^-22 of the product)
//   R S^T    rh sh + rh sl + rl sh      R^T A likewise
// 9 MFMAs per 32 x 32 x 16 step of the three contractions instead of 12, two S terms in LDS instead of three (64 KB), a
// third fewer operand reads in GEMM1.  Accumulators come out scaled by 2^(eA+eS) (P, undone before Y is subtracted),
// 2^(eR+eS) (gA) and 2^(eR+eA) (gSt), undone at the flushes; all of it exact.  Everything else is v7.
// ------------------------------------------------------------------------------------------------
constexpr int V8_SL_BYTES = 2 * V5_S_TERM;   // two terms per 32-column block
constexpr int V8_OFF_A = V5_NB * V8_SL_BYTES, V8_OFF_R = V8_OFF_A + V5_AIMG_BYTES, V8_LDS_BYTES = V8_OFF_R + 2 * V5_R_BYTES;
static_assert(V8_OFF_R % 256 == 0, "R images must start on a bank row");
static_assert(V8_LDS_BYTES <= 160 * 1024, "");

// LOSS: the sum of squares (nmf.py:13-25) is accumulated only by the instance the loss-only pass runs (doA = doS = 0:
// pmx_loglike, the backtracking line search); gradient passes never read it and skip its 16 multiply-adds per lane and block.
// R3 [r4]: the RESIDUAL to fp32's class.  With two terms per operand P = A S carries the operands' representation errors (2^-23 each): far
// below P's accumulation noise entry by entry, but COHERENT -- the same dS[k][n] in every row of P -- so the gradient contractions add
// them up over a whole column (gS picks up A^T A dS: M times, not sqrt(M) times, a single error) and they end up twice fp32's gradient
// error.  R3 adds the THIRD terms of A and S to the residual's product -- ah s3 + a3 sh beside ah sl + al sh -- and keeps everything but
// ah sh in a second accumulator, R = (P_hh - Y) + P_lo: what lies below half an ulp of P survives the cancellation.  Five MFMAs per k
// step instead of three in the producers, the consumers unchanged; S's third term in LDS (160 KB in all), A's in registers.
// scratch/r4_emulate_modes.py (NumPy, full cfg3): out-of-tolerance entries against the fp64 oracle 5.8 x / 4.2 x the fp32 oracle's with
// two terms (any number of terms in ONE accumulator: the same), 0.9 x / 1.0 x with this.
// HH [r5]: the residual from the HIGH x HIGH product alone (four MFMAs per block instead of twelve / twenty), P0 = a0 s0 with a0 = fp16(A 2^eA),
// s0 = fp16(S 2^eS) -- and what that leaves out restored EXACTLY, outside this kernel, through K x K matrices: A S - a0 s0 = A s_r + a_r s0
// (x_r = X - x0: exact in fp32), so  gA += A (s_r S^T) + a_r (s0 S^T),  gS += (A^T a_r) S + (A^T a0) s_r ... (k_gfix.hip: the correction arrives as
// one more gradient slab).  No representation error is left in P at all (mode f16x2r's third terms remove it to 2^-33), the residual's accumulation
// noise is exact fp32's, and the producers issue a third / a fifth of the MFMAs.  The consumers (two-term R, A, S: 3 products) are unchanged.
// scratch/r5_gradient_error_table.py: gradients 3.9e-8 / 1.5e-7 of max|g| against fp64 where NumPy fp32 has 3.6e-8 / 1.5e-7 and mode f16x2 4.0e-8 / 3.0e-7.
// RS [r5]: the consumers' ROLES split by contraction instead of by rows (gradient passes that want BOTH gradients).  Before: wave j owned rows 32 j of gA (both k
// tiles) and a quarter (row half x k tile) of gSt -- every wave read the block's S fragments (8 KB) AND its share of the panel's A fragments (8 KB) in every slot.
// <RS>: waves 0, 1 contract gA for 64 rows each (four accumulator tiles: the S fragments are read twice per slot instead of four times), waves 2, 3 contract
// gSt for one k tile each over ALL 128 rows of the panel -- with no gA tiles to hold, the panel's A fragments (64 registers) stay in registers for the eight
// slots of a panel.  24 MFMAs per wave and slot as before; LDS reads per slot 64 KB instead of 112 KB (profiles/r05_c_lds_ablation.txt priced those bytes);
// gSt needs no merge of row halves at the end.  Passes that want ONE gradient keep the row split (two of four waves would idle).
// ONLYS [r5]: the pass wants gSt alone (bsdmm's S step).  The row split's consumers then hold no gA tiles, and the 32 registers they leave keep each wave's share
// of the panel's A fragments for the panel's eight slots: 8 of the 16 KB a consumer reads per slot.
template <bool PROF, bool HASW, bool CHAIN, bool LOSS, bool R3 = false, bool HH = false, bool RS = false, bool ONLYS = false>
__global__ __launch_bounds__(V5_THREADS, 2) void k_grad_f16_v8(GradV4Args a) {
    static_assert(!ONLYS || (HH && !RS && !CHAIN && !PROF && !LOSS), "ONLYS: an instance of the <HH> gradient pass without gA");
    static_assert(!(R3 && HH) && !(HH && HASW), "HH: unweighted contexts, instead of the third terms");
    static_assert(!RS || (HH && !PROF && !LOSS), "RS: an instance of the <HH> gradient pass");
    constexpr int K = 64, ROWB = 128, NCB = V5_NB;
    constexpr int NT = R3 ? 3 : 2;                       // fp16 terms of A and S in the residual's product (HH: the second term serves the consumers only)
    constexpr int SLB = NT * V5_S_TERM, OFF_A = NCB * SLB, OFF_R = OFF_A + V5_AIMG_BYTES;
    static_assert(OFF_R + 2 * V5_R_BYTES <= 160 * 1024 && OFF_R % 256 == 0, "");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    if (chain_halted(a.status)) return;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lq = lane >> 4;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    int chainId = 0, chainPos = 0;           // CHAIN: which chain, and this workgroup's place in its rotation
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if constexpr (CHAIN) {
            // The chainL workgroups of a chain (same row region, consecutive column regions) are consecutive multiples of 8
            // apart in dispatch order, i.e. on ONE XCD where workgroup b runs on XCD b % 8 (checked at run time, below).
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
    // Column map: blocks 2q and 2q + 1 of a region share the 64 columns col0 + 64 q ..: block b's local column n is the
    // global column col0 + 64 (b >> 1) + 2 n + (b & 1) -- even columns to the even block, odd ones to the odd block.  A
    // producer lane then owns two ADJACENT columns (one per block) and one 8-byte load per row fetches both blocks' Y:
    // half the memory instructions, 256 contiguous bytes per row and instruction.  Only three places know the map: the S
    // staging below, the Y / W loads, the gSt flush.
    auto block_col = [&](int b, int n) { return col0 + 64 * (b >> 1) + 2 * n + (b & 1); };
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
    // CHAIN: the workgroup visits its row panels rotated by its place in the chain -- panel (t - chainPos) mod RP in the
    // t-th place -- so that at any time the members of a chain work on different panels, and member c reaches a panel one
    // panel-time after member c - 1 left it (every region has all RP panels in this mode: nrp == RP)
    auto panel_at = [&](int t) {
        if constexpr (CHAIN) { const int p = t - chainPos; return p < 0 ? p + nrp : p; }
        else return t;
    };

    if (T <= 0) {                              // region outside the matrix: its gSt slab part and loss partial are zero
        if (!producer && (j >> 1) == 0) {      // (one gSt slab per row region)
            const int kk = (j & 1) * 32 + l31;
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
            for (int c = 0; c < NCB; ++c)
                for (int i = 0; i < 16; ++i) {
                    const int gn = block_col(c, tile_row(i, lane));
                    if (gn < N && a.doS) dst[(int64_t)gn * K + kk] = 0.f;
                }
        }
        if (tid == 0) a.lossPart[blockIdx.x] = 0.0;
        return;
    }

    // [r5] the region's S rows are REQUESTED first (they do not depend on the scales): one dependent round trip less in front of the first block
    float4 sr[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)            // image row tid >> 4 of block c = S^T row block_col(c, tid >> 4)
        sr[c] = reinterpret_cast<const float4*>(a.St + (int64_t)block_col(c, tid >> 4) * K)[tid & 15];
    // ---- power-of-two operand scales from the factor maxima (k_absmax partials) and max|Y|; uniform ----------------
    float scA, scS, scR, unP, unA, unS;
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
        __syncthreads();                     // red aliases Sl
        if (f16_range_fault((float)K * mA * mS, a.ymax, a.rangeRatio, a.doA, a.doS, a.wstatus, tid)) return;
        int qA = 0, qS = 0, qR = 0;
        (void)frexpf(mA, &qA);               // m = f 2^q, f in [0.5, 1)  ->  m 2^(14-q) < 2^14
        (void)frexpf(mS, &qS);
        (void)frexpf((a.ymax + (float)K * mA * mS) * a.wmax, &qR);
        const int eA = mA > 0.f ? 14 - qA : 0, eS = mS > 0.f ? 14 - qS : 0, eR = 14 - qR;
        scA = ldexpf(1.f, eA); scS = ldexpf(1.f, eS); scR = ldexpf(1.f, eR);
        unP = ldexpf(1.f, -(eA + eS)); unA = ldexpf(1.f, -(eR + eS)); unS = ldexpf(1.f, -(eR + eA));
    }
    {   // ---- all S terms of the region, once: block cb -> Sl[cb] (all 512 threads, one float4 of each block) -------
        const int st_off = (tid >> 4) * ROWB + (((((tid & 15) >> 1) ^ v3_swz(tid >> 4)) & 7) << 4) + 8 * (tid & 1);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            unsigned char* d = smem + c * SLB + st_off;
            if constexpr (R3) {
                const float x[4] = {sr[c].x, sr[c].y, sr[c].z, sr[c].w};
                f16x4 t0, t1, t2;
#pragma unroll
                for (int q = 0; q < 4; ++q) { _Float16 h_, l_, m_; v8_split3(x[q], scS, h_, l_, m_); t0[q] = h_; t1[q] = l_; t2[q] = m_; }
                *reinterpret_cast<f16x4*>(d) = t0;
                *reinterpret_cast<f16x4*>(d + V5_S_TERM) = t1;
                *reinterpret_cast<f16x4*>(d + 2 * V5_S_TERM) = t2;
            } else {
                f16x4 t0, t1;
                v8_split2(sr[c], scS, t0, t1);
                *reinterpret_cast<f16x4*>(d) = t0;
                *reinterpret_cast<f16x4*>(d + V5_S_TERM) = t1;
            }
        }
    }

    if (producer) {
        k1_set_priority(-a.consPrio);      // (PMX_K1_PRIO < 0: the producers instead -- A/B only)
        // ================================ producers: GEMM1 and R =================================================
        f32x16 p0, p1;
        f32x16 q0, q1;                       // R3: the small products' accumulators (see the kernel's header)
        float yv[2][2][16];                  // Y in flight: [pair set][block of the pair][row i of the tile] (accumulator layout)
        float wv[2][HASW ? 16 : 1];          // weights of ONE block pair (requested a slot ahead of their first use: registers)
        float4 areg[4][2];
        f16x8 afr[4][NT];
        const int jw = __builtin_amdgcn_readfirstlane(j);
        // Y addresses: ONE wave-uniform base per request group (scalar registers: this wave's first row of the panel, the
        // block pair's first column) + sixteen per-lane byte offsets that never change (row i of the tile in the
        // accumulator's layout, this lane's column PAIR): the loads take the base as their scalar operand and no address
        // arithmetic is left in the loop (it was sixteen 64-bit vector adds per block).  The empty asm statements keep the
        // 32-bit offsets opaque: hipcc would otherwise widen them to 64 bits once, outside the loop, and add the base with
        // vector instructions again.  (The raw_buffer_load_b64 / _b128 builtins of this toolchain load ONE dword:
        // measured, not used.)  Nontemporal: Y is read once per launch; keeping it out of L2 / MALL leaves the gradient
        // slabs this kernel writes there for the update kernel that folds them (iteration -2.7 % at 16384 x 16384).
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        unsigned yoff[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) yoff[i] = ((unsigned)((i & 3) + 8 * (i >> 2) + 4 * hi) * (unsigned)a.ldY + 2u * (unsigned)l31) * 4u;
        const float* ybase0 = a.Y + (int64_t)(row0 + jw * 32) * a.ldY + col0;
        const float* wbase0 = HASW ? a.W + (int64_t)(row0 + jw * 32) * a.ldW + col0 : nullptr;   // (ldW == ldY: the launch checks it; W shares Y's offsets)
        // Y (and W) of the block pair q = blocks 2 q, 2 q + 1 (clamped past the end of the region) into pair set `set`:
        // rows i0 .. i0 + n - 1 of the sixteen (the requests of a pair are spread over the last MFMAs of a slot)
        auto pair_base = [&](int q, const float* b0, int64_t ld) {
            int brp = q >> 2;
            if (brp >= nrp) brp = nrp - 1;
            brp = panel_at(brp);
            return reinterpret_cast<const char*>(b0 + (int64_t)brp * V5_BM * ld + (q & 3) * 64);
        };
        auto load_pair_rows = [&](const char* base, const char* basew, auto set_c, auto i0_c, auto n_c) {
            constexpr int set = decltype(set_c)::value, i0 = decltype(i0_c)::value, n = decltype(n_c)::value;
#pragma unroll
            for (int i = i0; i < i0 + n; ++i) {
                asm volatile("" : "+v"(yoff[i]));
                const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(base + yoff[i]));
                yv[set][0][i] = v[0];
                yv[set][1][i] = v[1];
            }
            (void)basew;
        };
        auto load_w_rows = [&](const char* basew, auto i0_c, auto n_c) {      // the ONE set of weights
            constexpr int i0 = decltype(i0_c)::value, n = decltype(n_c)::value;
            if constexpr (HASW) {
#pragma unroll
                for (int i = i0; i < i0 + n; ++i) {
                    asm volatile("" : "+v"(yoff[i]));
                    const f32x2 u = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(basew + yoff[i]));
                    wv[0][i] = u[0];
                    wv[1][i] = u[1];
                }
            }
        };
        auto load_pair = [&](int q, auto set_c) {
            load_pair_rows(pair_base(q, ybase0, a.ldY), nullptr, set_c, std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
        };
        auto load_A = [&](int prow) {
            const float4* src = reinterpret_cast<const float4*>(a.A + (int64_t)(prow + j * 32 + l31) * K + hi * 8);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                areg[ks][0] = src[ks * 4];
                areg[ks][1] = src[ks * 4 + 1];
            }
        };
        auto make_afr = [&]() {              // split the scaled panel rows into two fp16 terms (register fragments of GEMM1's A operand)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float x[8] = {areg[ks][0].x, areg[ks][0].y, areg[ks][0].z, areg[ks][0].w,
                                    areg[ks][1].x, areg[ks][1].y, areg[ks][1].z, areg[ks][1].w};
                if constexpr (R3) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { _Float16 h_, l_, m_; v8_split3(x[q], scA, h_, l_, m_); afr[ks][0][q] = h_; afr[ks][1][q] = l_; afr[ks][2][q] = m_; }
                } else {
                unsigned hh[4], ll[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v8_split_pair(x[2 * q], x[2 * q + 1], scA, hh[q], ll[q]);
                afr[ks][0] = __builtin_bit_cast(f16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
                afr[ks][1] = __builtin_bit_cast(f16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
                }
            }
        };
        const int pa0 = (j * 32 + l31) * ROWB + ((hi ^ v3_swz(j * 32 + l31)) << 4);   // chunk 2 ks + hi: ^ (ks << 5)
        auto publish_A = [&]() {             // terms 0,1 of the current panel -> Aimg, for the consumers' gSt contraction
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                *reinterpret_cast<f16x8*>(smem + OFF_A + (pa0 ^ (ks << 5))) = afr[ks][0];
                *reinterpret_cast<f16x8*>(smem + OFF_A + V5_A_TERM + (pa0 ^ (ks << 5))) = afr[ks][1];
            }
        };
        const int s_g1 = l31 * ROWB + ((hi ^ v3_swz(l31)) << 4);                 // GEMM1 B operand: row l31, chunk 2 ks + hi: ^ (ks << 5)
        const int r_w = l31 * 256 + (((4 * j) ^ v4_swz(l31)) << 4) + 8 * hi;      // R producer, ^ (g << 4)
        using yes = std::integral_constant<bool, true>;
        using no = std::integral_constant<bool, false>;
        using set0 = std::integral_constant<int, 0>;
        using set1 = std::integral_constant<int, 1>;
        if constexpr (!HASW && !R3) load_A(row0 + panel_at(0) * V5_BM);
        load_pair(0, set0{});                // slot s (even) requests the pair of blocks s + 2, s + 3 into the set block s - 1 has just left
        // slot 0 runs the same code as every other slot (no peeled copy: the loop head then sees the same requests in flight
        // from both sides and the compiler's wait counts stay exact): its epilogue works on a zero "block -1" -- R = 0 into
        // an image nobody reads before block 1 rewrites it, nothing added to the loss -- and requests the pair of blocks 2, 3
#pragma unroll
        for (int i = 0; i < 16; ++i) { p1[i] = 0.f; q1[i] = 0.f; q0[i] = 0.f; yv[1][1][i] = 0.f; if constexpr (HASW) { wv[0][i] = 0.f; wv[1][i] = 0.f; } }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();        // Sl published

        // One slot; cb (the block's place in its panel) is a compile-time constant: the eight slots of a panel are ONE basic
        // block, every LDS address is a register plus an immediate, and the compiler counts the loads in flight exactly.
        // GEMM: block s = (rp, cb) into pc.  EPI: block s - 1 from pp and its Y tile -> R[(s - 1) & 1].
        auto slot = [&](int rp, auto cb_c, f32x16& pc, f32x16& pp, f32x16& lc, f32x16& lp, auto gemm_c, auto epi_c) {
            constexpr int cb = decltype(cb_c)::value;
            constexpr bool GEMM = decltype(gemm_c)::value, EPI = decltype(epi_c)::value;
            constexpr int pset = ((cb + 7) >> 1) & 1, ptile = (cb + 7) & 1;     // pair set and place in its pair of block s - 1 (8 blocks per panel)
            if constexpr (cb == 2 && GEMM) {         // block s-2 opened this row panel: the consumers start on it in this slot
                publish_A();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
            PH(5)
            if constexpr (GEMM && cb == 0) { // block s opens a row panel: its A terms (rows requested 8 slots ago), then the next panel's rows
                if constexpr (HASW || R3) {  // (weighted: the registers of that prefetch hold weights -- R3: the second accumulator and the third terms --; the rows are fetched here, an L2 trip per panel)
                    load_A(row0 + panel_at(rp) * V5_BM);
                    make_afr();
                } else {
                    make_afr();
                    load_A(row0 + panel_at(rp + 1 < nrp ? rp + 1 : nrp - 1) * V5_BM);
                }
            }
            f16x8 sv[4][NT];
            const unsigned char* Slb = smem + cb * SLB;
            auto read_sv = [&](int ks) {     // R3: the fragments of k step ks, requested one k step (five MFMAs) ahead of their first reader
                const int so = s_g1 ^ (ks << 5);
                sv[ks][0] = *reinterpret_cast<const f16x8*>(Slb + so);
                sv[ks][1] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                if constexpr (R3) sv[ks][2] = *reinterpret_cast<const f16x8*>(Slb + so + 2 * V5_S_TERM);
            };
            if constexpr (GEMM && R3) {
                read_sv(0);
            } else if constexpr (GEMM) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int so = s_g1 ^ (ks << 5);
                    sv[ks][0] = *reinterpret_cast<const f16x8*>(Slb + so);
                    if constexpr (!HH) sv[ks][1] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                    if constexpr (R3) sv[ks][2] = *reinterpret_cast<const f16x8*>(Slb + so + 2 * V5_S_TERM);
                }
            }
            PH(2)
            // The slot's instruction order is laid out by hand, twelve steps of ONE MFMA of block s (a dependent chain: the
            // wave would sit behind each of them for 32 cycles) + a piece of block s - 1's epilogue that issues in its shadow:
            // steps 0-7 one pair of R entries each (residual, two-term split: six vector instructions; every second step the
            // two 8-byte stores of a finished group), steps 8-11 four of the sixteen requests of the next block pair.  The
            // fences keep hipcc from regrouping it (left alone it ran the whole epilogue first and the twelve MFMAs after it).
            unsigned char* Rb = smem + OFF_R + ((cb + 1) & 1) * V5_R_BYTES;    // block s - 1 has the other parity
            unsigned h2[4][2], l2[4][2];
            const char* ybase_n = nullptr;
            const char* wbase_n = nullptr;
            if constexpr (EPI && (cb & 1) == 0) {    // the pair set of block s - 1 (the second of its pair) is free after this epilogue: blocks s + 2, s + 3
                ybase_n = pair_base(rp * 4 + (cb >> 1) + 1, ybase0, a.ldY);
                if constexpr (HASW) wbase_n = pair_base(rp * 4 + (cb >> 1), wbase0, a.ldW);   // blocks s, s + 1: their epilogues are the next two slots
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < (R3 ? 20 : 12); ++t) {
                if constexpr (GEMM && R3) {
                    // per k step: al sh, ah sl, a3 sh, ah s3 into the small products' accumulator, ah sh into the other (steps 12-19 have
                    // no epilogue piece beside them: the epilogue is eight pieces and four groups of requests as before)
                    const int ks = t / 5, wh = t % 5;
                    if (wh == 0 && ks < 3) read_sv(ks + 1);
                    if (wh == 4) {
                        f32x16 cin = pc;
                        if (t == 4) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                        }
                        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sv[ks][0], cin, 0, 0, 0);
                    } else {
                        f32x16 cin = lc;
                        if (t == 0) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                        }
                        lc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][wh == 0 ? 1 : (wh == 2 ? 2 : 0)], sv[ks][wh == 1 ? 1 : (wh == 3 ? 2 : 0)], cin, 0, 0, 0);
                    }
                } else if constexpr (GEMM && HH) {
                    // ah sh alone: one MFMA every third step, the epilogue pieces between them.  (Measured and dropped: every k step from C = 0
                    // with the partial sums added by the VALU, and the odd k steps with flipped sign -- the fp16 MFMA's adder is not IEEE
                    // (scratch/r5_mfma_rounding2.hip), but neither changes the gradients' error: profiles/r05_b_mfma_accumulation.txt)
                    if (t % 3 == 0) {
                        f32x16 cin = pc;
                        if (t == 0) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                        }
                        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[t / 3][0], sv[t / 3][0], cin, 0, 0, 0);
                    }
                } else if constexpr (GEMM) {
                    const int ks = t / 3, wh = t % 3;        // al sh, ah sl, ah sh
                    f32x16 cin = pc;
                    if (t == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                    }
                    pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][wh == 0 ? 1 : 0], sv[ks][wh == 1 ? 1 : 0], cin, 0, 0, 0);
                }
                if constexpr (EPI) {
                    if (t < 8) {
                        const int g = t >> 1, hf = t & 1;
                        float r[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int e = 4 * g + 2 * hf + q;
                            r[q] = pp[e] * unP - yv[pset][ptile][e];
                            if constexpr (R3) r[q] = __builtin_fmaf(lp[e], unP, r[q]);     // (P_hh - Y) + P_lo
                            if constexpr (HASW) {
                                const float ww = wv[ptile][e];
                                if constexpr (LOSS) lossAcc += ww * (r[q] * r[q]);
                                r[q] *= ww;
                            } else {
                                if constexpr (LOSS) lossAcc += r[q] * r[q];
                            }
                        }
                        v8_split_pair(r[0], r[1], scR, h2[g][hf], l2[g][hf]);
                        if (hf == 1) {
                            const int o = r_w ^ (g << 4);
                            *reinterpret_cast<uint2*>(Rb + o) = make_uint2(h2[g][0], h2[g][1]);
                            *reinterpret_cast<uint2*>(Rb + V5_R_TERM + o) = make_uint2(l2[g][0], l2[g][1]);
                        }
                    } else if constexpr ((cb & 1) == 0) {
                        if (t >= 12) {}
                        else
                        if (t == 8) load_pair_rows(ybase_n, wbase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
                        if (t == 9) load_pair_rows(ybase_n, wbase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
                        if (t == 10) load_pair_rows(ybase_n, wbase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
                        if (t == 11) load_pair_rows(ybase_n, wbase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
                        if constexpr (HASW) {    // weights of the pair whose first block's epilogue is the NEXT slot's
                            if (t == 8) load_w_rows(wbase_n, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
                            if (t == 10) load_w_rows(wbase_n, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PH(3)
            PH(4)
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            __builtin_amdgcn_s_barrier();
            PH(0)
        };
        using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
        using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
        using c4 = std::integral_constant<int, 4>; using c5 = std::integral_constant<int, 5>;
        using c6 = std::integral_constant<int, 6>; using c7 = std::integral_constant<int, 7>;
        // even blocks: accumulator p0; odd blocks: p1
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            slot(rp, c0{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c1{}, p1, p0, q1, q0, yes{}, yes{});
            slot(rp, c2{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c3{}, p1, p0, q1, q0, yes{}, yes{});
            slot(rp, c4{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c5{}, p1, p0, q1, q0, yes{}, yes{});
            slot(rp, c6{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c7{}, p1, p0, q1, q0, yes{}, yes{});
        }
        slot(nrp, c0{}, p0, p1, q0, q1, no{}, yes{});
        slot(nrp, c1{}, p1, p0, q1, q0, no{}, no{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (pairs with the consumers' barrier between parking and merging their gSt row halves)
    } else if constexpr (RS) {
        // ================================ consumers, roles split by contraction (see RS in the header) ==============
        k1_set_priority(a.consPrio);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // Sl published
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        };
        auto tr_src = [&](int row, int k0) {
            const int kk = k0 + 16 * (lq & 1) + 4 * (li & 3);
            return row * ROWB + ((((kk >> 3) ^ v3_swz(row)) & 7) << 4) + 8 * ((kk >> 2) & 1);
        };
        if (j < 2) {
            // ---- gA: rows 64 j .. 64 j + 63 of the panel (row tiles rt = 0, 1), both k tiles --------------------------------
            f32x16 accA[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) { accA[rt][0][i] = 0.f; accA[rt][1][i] = 0.f; }
            int r_t[2][2];                       // R as the A operand (transposing read), per row tile
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int m = (2 * j + rt) * 32 + 16 * (lq & 1) + 4 * (li & 3);
                const int n0 = 8 * hi + (li >> 2), n1 = n0 + 4;
                r_t[rt][0] = n0 * 256 + ((((m >> 3) ^ v4_swz(n0)) & 15) << 4) + 8 * ((m >> 2) & 1);
                r_t[rt][1] = n1 * 256 + ((((m >> 3) ^ v4_swz(n1)) & 15) << 4) + 8 * ((m >> 2) & 1);
            }
            const int s_t0 = tr_src(8 * hi + (li >> 2), 0), s_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);   // S as the B operand; k tile 1: ^ 64
            const int slabIdxA = CHAIN ? colRegion / a.chainL : colRegion;
            auto gA_tile = [&](int prow, int rt) { return a.slabA + (int64_t)slabIdxA * M * K + (int64_t)(prow + (2 * j + rt) * 32 + 4 * hi) * K + l31; };
            auto flush_gA = [&](int prow) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float* p0_ = gA_tile(prow, rt);
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float* ph_ = p0_ + half * 16 * K;
                        asm volatile("" : "+v"(ph_));
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int i = half * 8 + q;        // tile_row(i) = (i & 3) + 8 * (i >> 2) + 4 * hi
                            const int ro = ((q & 3) + 8 * (q >> 2)) * K;
                            ph_[ro] = accA[rt][0][i] * unA;
                            ph_[ro + 32] = accA[rt][1][i] * unA;
                        }
                    }
                }
            };
            const float invUnA = scR * scS;
            ChainLink link;
            if constexpr (CHAIN) link.init(a.chainFlags, chainId, nrp, j, a.status, a.wstatus, lane);
            if constexpr (CHAIN) {
                if (a.chainInject && blockIdx.x == 0 && j == 0) link.fault(3);
            }
            sync();
            sync();
            int s = 2;
            // [r6] The previous sum of the panel arrives in four pieces (accumulator registers 4 p .. 4 p + 3 of all four tiles), piece p added behind the MFMAs of
            // slot 4 + p.  Requested in FRONT of those MFMAs (rounds 2-5) a piece had 24 MFMAs = ~0.4 us to arrive -- less than a round trip to the L2 under the
            // stream of Y, and the consumers' pole waited for the rest in four slots of every panel (the 6 % the chains cost inside K1, profiles/r05_d_chain_length.txt).
            // Now the pieces are requested PF at a time as soon as the arrival word has been seen (behind slot 3's MFMAs) and piece p + PF behind the add of piece p:
            // a whole slot (barrier, operand reads, MFMAs) or more per round trip.  The adds stand where they stood: the same sums in the same order, bit for bit.
            constexpr int PF = PMX_CHAIN_PF;          // pieces in flight (buffers of 16 registers): 0 = the old placement
            static_assert(PF >= 0 && PF <= 4, "");
            float pv[PF > 0 ? PF : 1][2][2][4];
            auto fetch_piece = [&](int prow_, auto p_c) {
                constexpr int p = decltype(p_c)::value;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const float* pb = gA_tile(prow_, rt) + (8 * (p & 1) + 16 * (p >> 1)) * K;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pv[PF > 0 ? p % PF : 0][rt][0][q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + q * K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        pv[PF > 0 ? p % PF : 0][rt][1][q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)(pb + q * K + 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
                }
            };
#pragma nounroll
            for (int rp = 0; rp < nrp; ++rp) {
                const int pnl = panel_at(rp);
                const int prow = row0 + pnl * V5_BM;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if (cb == 0) {
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_s_barrier();
                        if constexpr (CHAIN) link.open(pnl, chainPos, a.chainL, 1, nrp, a.chainBase, (a.doA & 1) != 0);
                    }
                    if constexpr (CHAIN) {
                        if (cb == 3) link.look();
                        if constexpr (PF == 0) {
                            if (cb >= 4 && link.cadd) {
                                if (cb == 4) fetch_piece(prow, std::integral_constant<int, 0>{});
                                if (cb == 5) fetch_piece(prow, std::integral_constant<int, 1>{});
                                if (cb == 6) fetch_piece(prow, std::integral_constant<int, 2>{});
                                if (cb == 7) fetch_piece(prow, std::integral_constant<int, 3>{});
                            }
                        }
                    }
                    {
                        const unsigned char* Rb = smem + OFF_R + ((s - 2) & 1) * V5_R_BYTES;
                        const unsigned char* Slb = smem + cb * SLB;
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const int so0 = s_t0 + ks * 16 * ROWB, so1 = s_t1 + ks * 16 * ROWB;
                            const f16x8 s00 = v8_tr_pair(Slb, so0, so1);
                            const f16x8 s01 = v8_tr_pair(Slb + V5_S_TERM, so0, so1);
                            const f16x8 s10 = v8_tr_pair(Slb, so0 ^ 64, so1 ^ 64);
                            const f16x8 s11 = v8_tr_pair(Slb + V5_S_TERM, so0 ^ 64, so1 ^ 64);
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt) {
                                const f16x8 r0 = v8_tr_pair(Rb, r_t[rt][0] + ks * 4096, r_t[rt][1] + ks * 4096);
                                const f16x8 r1 = v8_tr_pair(Rb + V5_R_TERM, r_t[rt][0] + ks * 4096, r_t[rt][1] + ks * 4096);
                                accA[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s00, accA[rt][0], 0, 0, 0);
                                accA[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s10, accA[rt][1], 0, 0, 0);
                                accA[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s01, accA[rt][0], 0, 0, 0);
                                accA[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s11, accA[rt][1], 0, 0, 0);
                                accA[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s00, accA[rt][0], 0, 0, 0);
                                accA[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s10, accA[rt][1], 0, 0, 0);
                            }
                        }
                    }
                    if constexpr (CHAIN) {
                        if (cb == 3) {
                            link.wait();
                            if constexpr (PF > 0) {
                                if (link.cadd) {
                                    fetch_piece(prow, std::integral_constant<int, 0>{});
                                    if constexpr (PF > 1) fetch_piece(prow, std::integral_constant<int, 1>{});
                                    if constexpr (PF > 2) fetch_piece(prow, std::integral_constant<int, 2>{});
                                    if constexpr (PF > 3) fetch_piece(prow, std::integral_constant<int, 3>{});
                                }
                            }
                        }
                        if (cb >= 4 && link.cadd) {
                            const int b_ = PF > 0 ? (cb - 4) % PF : 0;
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    accA[rt][0][4 * (cb - 4) + q] += pv[b_][rt][0][q] * invUnA;
                                    accA[rt][1][4 * (cb - 4) + q] += pv[b_][rt][1][q] * invUnA;
                                }
                            if constexpr (PF > 0 && PF < 4) {          // the buffer is free: the piece PF places on
                                if (cb - 4 + PF == 1) fetch_piece(prow, std::integral_constant<int, 1>{});
                                if (cb - 4 + PF == 2) fetch_piece(prow, std::integral_constant<int, 2>{});
                                if (cb - 4 + PF == 3) fetch_piece(prow, std::integral_constant<int, 3>{});
                            }
                        }
                    }
                    if (cb + 1 == NCB) {
                        flush_gA(prow);
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int i = 0; i < 16; ++i) { accA[rt][0][i] = 0.f; accA[rt][1][i] = 0.f; }
                        if constexpr (CHAIN) link.flushed();
                    }
                    sync();
                    ++s;
                }
            }
            if constexpr (CHAIN) link.publish();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();          // (the producers' last barrier)
        } else {
            // ---- gSt: k tile kt of every block of the region, ALL 128 rows of a panel (eight steps of sixteen) ------------------
            const int kt = j - 2;
            f32x16 accS[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) accS[c][i] = 0.f;
            const int r_g = l31 * 256 + ((hi ^ v4_swz(l31)) << 4);                                   // R^T rows n as the A operand, rows 16 ks ..: ^ (ks << 5)
            const int a_t0 = tr_src(8 * hi + (li >> 2), kt * 32), a_t1 = tr_src(8 * hi + 4 + (li >> 2), kt * 32);   // A as the B operand, rows 16 ks ..: + ks * 16 * ROWB
            f16x8 af[8][2];                      // the panel's A fragments (both terms), fetched once per panel
            sync();
            sync();
            int s = 2;
#pragma nounroll
            for (int rp = 0; rp < nrp; ++rp) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if (cb == 0) {               // block s-2 opens a row panel: the producers have just published its A terms
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_s_barrier();
                        const unsigned char* Ab = smem + OFF_A;
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) {
                            af[ks][0] = v8_tr_pair(Ab, a_t0 + ks * 16 * ROWB, a_t1 + ks * 16 * ROWB);
                            af[ks][1] = v8_tr_pair(Ab + V5_A_TERM, a_t0 + ks * 16 * ROWB, a_t1 + ks * 16 * ROWB);
                        }
                    }
                    const unsigned char* Rb = smem + OFF_R + ((s - 2) & 1) * V5_R_BYTES;
                    // The panel's contribution is summed in a FRESH accumulator and added to the launch-long one with an IEEE fp32 add: the fp16 MFMA cuts its
                    // addends three bits below the last place of the largest (profiles/r05_b_mfma_accumulation.txt) -- 24 truncating steps per slot on an
                    // accumulator that has grown over the whole region drift downwards; on a small one they do not, and the add rounds to nearest once per slot.
                    f32x16 tacc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) tacc[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int ro = r_g ^ (ks << 5);
                        const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                        const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
                        tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, af[ks][0], tacc, 0, 0, 0);
                        tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, af[ks][1], tacc, 0, 0, 0);
                        tacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, af[ks][0], tacc, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) accS[cb][i] += tacc[i];
                    sync();
                    ++s;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();          // (the producers' last barrier)
            {
                float* dst = a.slabS + (int64_t)rowRegion * N * K;
                const int kk = kt * 32 + l31;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int gn = block_col(cb, tile_row(i, lane));
                        dst[(int64_t)gn * K + kk] = accS[cb][i] * unS;
                    }
            }
        }
    } else {
        // ================================ consumers: GEMM2 and GEMM3 of block s-2 =================================
        k1_set_priority(a.consPrio);
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
        // gA slab this workgroup contributes to: its own (one per column region), or its chain's (accumulated in place)
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
                    ph_[ro] = accA0[i] * unA;
                    ph_[ro + 32] = accA1[i] * unA;
                }
            }
        };
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            PH(9)
        };
        f16x8 afS[ONLYS ? 4 : 1][2];        // ONLYS: this wave's share (row half mh, k tile kt) of the panel's A fragments
        auto consume = [&](int b, int prow, int cb, f32x16& accSc) {     // block b: column block cb of the panel at row prow
            const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
            const unsigned char* Slb = smem + cb * SLB;
            const unsigned char* Ab = smem + OFF_A;
            if (!ONLYS && (a.doA & 1)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
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
            }
            PH(6)
            if (a.doS) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
                    f16x8 a0, a1;
                    if constexpr (ONLYS) { a0 = afS[ks][0]; a1 = afS[ks][1]; }
                    else { a0 = v8_tr_pair(Ab, ao0, ao1); a1 = v8_tr_pair(Ab + V5_A_TERM, ao0, ao1); }
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, accSc, 0, 0, 0);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, accSc, 0, 0, 0);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, accSc, 0, 0, 0);
                }
            }
            if (!ONLYS && !CHAIN && (a.doA & 1) && cb + 1 == NCB) {
                flush_gA(prow);
#pragma unroll
                for (int i = 0; i < 16; ++i) { accA0[i] = 0.f; accA1[i] = 0.f; }
            }
        };
        // ---- CHAIN: gA summed in place, through the XCD's L2 (chain_link.h: protocol, fault handling) --------------------
        // The previous sum is fetched in four pieces during the panel's last four column blocks (requested before a block's
        // MFMAs, added after them), the arrival word of a finished panel is published one slot later: no wait of the
        // protocol sits in front of work.
        const float invUnA = scR * scS;              // 2^(eR+eS): previous sums enter the accumulators in their scale
        ChainLink link;
        if constexpr (CHAIN) link.init(a.chainFlags, chainId, nrp, j, a.status, a.wstatus, lane);
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
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (cb == 0) {               // block s-2 opens a row panel: the producers publish its A terms now
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    if constexpr (CHAIN) link.open(pnl, chainPos, a.chainL, 1, nrp, a.chainBase, (a.doA & 1) != 0);
                    if constexpr (ONLYS) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            afS[ks][0] = v8_tr_pair(smem + OFF_A, a_t0 + ks * 16 * ROWB, a_t1 + ks * 16 * ROWB);
                            afS[ks][1] = v8_tr_pair(smem + OFF_A + V5_A_TERM, a_t0 + ks * 16 * ROWB, a_t1 + ks * 16 * ROWB);
                        }
                    }
                }
                float pv0[4], pv1[4];
                if constexpr (CHAIN) {
                    if (cb == 3) link.look();
                    if (cb >= 4 && link.cadd) {   // piece cb - 4 of the previous sum: accumulator registers 4 (cb - 4) ..
                        const float* pb = gA_tile(prow) + (8 * ((cb - 4) & 1) + 16 * ((cb - 4) >> 1)) * K;
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
                            accA0[4 * (cb - 4) + q] += pv0[q] * invUnA;
                            accA1[4 * (cb - 4) + q] += pv1[q] * invUnA;
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
        // [r4] ONE gSt slab per row region: the two row halves of a tile (waves mh = 0 / 1 of a k tile) are summed here through the
        // launch's own LDS (every image is dead behind the loop's last barrier): each wave parks the four blocks the OTHER half
        // finishes and adds the other's to its own four -- half the stores of rounds 1-3 (which wrote two slabs per row region and
        // left the sum to the update kernel: 8 slabs = 32 MB to fold at cfg3, a third of the adaprox tail's traffic)
        {
            constexpr int CB = NCB / 2;
            float* fsm = reinterpret_cast<float*>(smem);
            auto halves = [&](auto MH) {
                constexpr int m = decltype(MH)::value;
                float* park = fsm + ((kt * 2 + m) * CB) * 1024 + lane;               // [kt][writer][block][i][lane]
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
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int gn = block_col(m * CB + cc, tile_row(i, lane));
                            const float o = oth[cc * 1024 + i * 64], own = accS[m * CB + cc][i];
                            dst[(int64_t)gn * K + kk] = (m == 0 ? own + o : o + own) * unS;        // rows 0-63 + rows 64-127
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

template <bool PROF, bool HASW, bool CHAIN, bool LOSS, bool R3 = false, bool HH = false, bool RS = false, bool ONLYS = false>
static hipError_t grad_launch_f16_v8_t(const GradV4Args& a, hipStream_t stream) {
    constexpr int lds = R3 ? V7_LDS_BYTES : V8_LDS_BYTES;        // (three S terms: the split-bf16 kernel's 160 KB)
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f16_v8<PROF, HASW, CHAIN, LOSS, R3, HH, RS, ONLYS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f16_v8<PROF, HASW, CHAIN, LOSS, R3, HH, RS, ONLYS>), dim3(a.gridX * a.gridY), dim3(V5_THREADS), lds, stream, a);
    return hipGetLastError();
}
static hipError_t grad_launch_f16_v8(const GradV4Args& a, hipStream_t stream) {
    // r3 == 2 [r5]: the high x high residual whose missing terms arrive as a correction slab (<HH>; gradient passes only -- the loss-only pass
    // has nowhere to put a correction and runs the third terms' instance)
    if (a.r3 == 2 && a.W == nullptr && (a.doA & 1) && a.doS && !a.prof && !(getenv("PMX_K1_ROLE_SPLIT") && atoi(getenv("PMX_K1_ROLE_SPLIT")) == 0))   // <RS>: both gradients wanted (PMX_K1_ROLE_SPLIT=0: A/B)
        return a.chainL > 0 ? grad_launch_f16_v8_t<false, false, true, false, false, true, true>(a, stream) : grad_launch_f16_v8_t<false, false, false, false, false, true, true>(a, stream);
    if (a.r3 == 2 && a.W == nullptr && !(a.doA & 1) && a.doS && !(getenv("PMX_K1_ROLE_SPLIT") && atoi(getenv("PMX_K1_ROLE_SPLIT")) == 0))      // <ONLYS>: gSt alone (same A/B switch)
        return grad_launch_f16_v8_t<false, false, false, false, false, true, false, true>(a, stream);
    if (a.r3 == 2 && a.W == nullptr && ((a.doA & 1) || a.doS))
        return a.chainL > 0 ? grad_launch_f16_v8_t<false, false, true, false, false, true>(a, stream) : grad_launch_f16_v8_t<false, false, false, false, false, true>(a, stream);
    if (a.r3 && a.W == nullptr) {    // R3: the residual to fp32's class (unweighted instances; a weighted context keeps two terms)
        if (!(a.doA & 1) && !a.doS) return grad_launch_f16_v8_t<false, false, false, true, true>(a, stream);
        return a.chainL > 0 ? grad_launch_f16_v8_t<false, false, true, false, true>(a, stream) : grad_launch_f16_v8_t<false, false, false, false, true>(a, stream);
    }
    if (!(a.doA & 1) && !a.doS)      // the loss-only pass (no gradient is written: nothing to chain)
        return a.W != nullptr ? grad_launch_f16_v8_t<false, true, false, true>(a, stream) : grad_launch_f16_v8_t<false, false, false, true>(a, stream);
    if (a.chainL > 0) return a.W != nullptr ? grad_launch_f16_v8_t<false, true, true, false>(a, stream) : grad_launch_f16_v8_t<false, false, true, false>(a, stream);
    if (a.W != nullptr) return grad_launch_f16_v8_t<false, true, false, false>(a, stream);   // (no phase profiling of the weighted instance)
    return a.prof ? grad_launch_f16_v8_t<true, false, false, false>(a, stream) : grad_launch_f16_v8_t<false, false, false, false>(a, stream);
}
