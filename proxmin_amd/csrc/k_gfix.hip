// [r5] The correction that lets K1 form its residual from ONE fp16 product (k_grad_f16_v8<.., HH>, k_grad_f16_k128<.., HH>).
//
// Reference: nmf.grad_likelihood (proxmin/nmf.py:28-41)  D = A S - Y,  gA = D S^T,  gS = A^T D  in the caller's fp32.
// K1 <HH> computes P0 = a0 s0 with a0 = fp16(A 2^eA) 2^-eA, s0 = fp16(S 2^eS) 2^-eS (11 significant bits each, products exact,
// fp32 accumulation) and contracts R0 = P0 - Y.  What the high terms leave out is known exactly,
//     A S - a0 s0  =  A s_r + a_r s0  =  a0 s_r + a_r S          (x_r = X - x0: an exact fp32 subtraction),
// and its contribution to both gradients goes through K x K matrices instead of M x N x K products:
//     gA  += A  (s_r S^T) + a_r (s0 S^T)             gSt += St (a_r^T A) + st_r (a0^T A)
// i.e. for either block X with the OTHER factor Z (both tall, rows x K):   C_X = X Qr(Z) + x_r Q0(Z),
//     Qr(Z) = z_r^T Z,   Q0(Z) = z0^T Z              (K x K, Q[k'][k] = sum over rows of U[row][k'] Z[row][k]).
// 4 (M + N) K^2 flops beside K1's 6 M N K: 0.1 % at 16384^2 x 64.  The correction is 2^-12 of the gradient's terms and needs 2^-12 of
// relative accuracy to leave exact fp32's error class untouched; it is computed to ~2^-22.  Three launches behind K1 (they depend on the factors only):
//   k_gfix_gram    per-workgroup partial Q0 / Qr of both factors (fp16 MFMA on the very terms K1 uses: z0 exact, Z = z0 + z1, z_r ~ z1)
//   k_gfix_reduce  fixed-order sum of the partials (deterministic: no float atomics)
//   k_gfix_apply   C_X for every row of both blocks (split-fp16 MFMA: the rows as three fp16 terms, the matrices as two, transposed in LDS) into the
//                  CORRECTION SLAB the update kernels fold behind K1's own slabs (SlabRef::extra)
// Unweighted likelihood only: with weights the missing term is W o (A s_r + a_r s0), which does not factor.
// Row-sharded runs: every term is a sum over the rank's own rows (A's) or over replicated data (S's) -- nothing to exchange.
#include "pmx_common.h"

constexpr int GFIX_PARTS = 128;          // partial matrices per factor (one workgroup each)

struct GfixArgs {
    const float* X[2];       // K1's operands: A (M x K), St (N x K), row pitch K (K1's K: 64 or 128)
    int64_t rows[2];
    int K;
    const float* absmax;     // [2][256] partial maxima of |A|, |St|: the operand scales are K1's own (k_absmax / the update kernels' finish)
    float* part;             // [2][GFIX_PARTS][2][K * K]
    float* Q;                // [2][2][K * K]: Q[f][0] = x0^T X, Q[f][1] = x_r^T X of factor f
    float* out[2];           // correction slab of block j (rows x ld)
    int ld;
    int want[2];             // correct gA / gSt
    const DevStatus* status;
    long long* prof;         // tuning (PMX_GFIX_PROF=1): 100 MHz time stamps of workgroup (0, 0): [0..3] gram, [4..5] reduce, [6..10] apply; else nullptr
};
#define GFIX_STAMP(i) do { if (a.prof != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.prof[i] = wall_clock64(); } while (0)

typedef _Float16 gf16x8 __attribute__((ext_vector_type(8)));

// 2^e with max|X_f| 2^e in [2^13, 2^14): K1's scale (k_grad_f16_v8: eA / eS)
__device__ __forceinline__ float gfix_scale(const float* absmax, int f, float* red) {
    float m = 0.f;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) m = fmaxf(m, absmax[f * 256 + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    float mx = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, red[i]);
    int q = 0;
    (void)frexpf(mx, &q);
    return ldexpf(1.f, mx > 0.f ? 14 - q : 0);
}

// KT = K / 32.  Four waves; wave w owns the output tiles (ti, tj) of linear index w KT^2 / 4 .. (KT = 2: one tile, KT = 4: a tile row)
// of both matrices and walks ALL rows of the workgroup's share, sixteen at a time (one 32 x 32 x 16 step).  The kernel is a chain of
// memory round trips, not arithmetic (a share is 128 rows at 16384): the loads of NB steps are in flight together, and the first batch
// goes out before the scale's own round trip.
template <int KT, int W>
__device__ __forceinline__ void gfix_gram_wave(const GfixArgs& a, int f, float* red) {
    constexpr int K = 32 * KT, TPW = KT * KT / 4, NB = KT == 2 ? 8 : 4;    // (K = 64: a 128-row share is ONE batch of loads, K = 128: two)
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int64_t rows = a.rows[f];
    const int64_t per = 16 * ((rows + 16 * GFIX_PARTS - 1) / (16 * GFIX_PARTS));
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float* X = a.X[f] + 32 * 0 + l31;
    f32x16 acc0[TPW], accr[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[t][i] = 0.f; accr[t][i] = 0.f; }
    float v[NB][KT][8];
#define GFIX_REQUEST(rb_)                                                                          \
    _Pragma("unroll") for (int s = 0; s < NB; ++s)                                                 \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                \
        const int64_t r = (rb_) + 16 * s + 8 * hi + q;                                             \
        _Pragma("unroll") for (int c = 0; c < KT; ++c) v[s][c][q] = r < r1 ? X[r * K + 32 * c] : 0.f; \
    }
    GFIX_STAMP(0);
    if (r0 < r1) { GFIX_REQUEST(r0) }
    const float sc = gfix_scale(a.absmax, f, red), un = 1.f / (sc * sc);
    GFIX_STAMP(1);
    for (int64_t rb = r0; rb < r1; rb += 16 * NB) {
        gf16x8 h[NB][KT], l[NB][KT];
#pragma unroll
        for (int s = 0; s < NB; ++s)
#pragma unroll
            for (int c = 0; c < KT; ++c)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float x = v[s][c][q] * sc;
                    const _Float16 t = (_Float16)x;
                    h[s][c][q] = t;
                    l[s][c][q] = (_Float16)(x - (float)t);
                }
        if (rb + 16 * NB < r1) { GFIX_REQUEST(rb + 16 * NB) }
#pragma unroll
        for (int s = 0; s < NB; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                constexpr int idx0 = W * TPW;
                const int ti = (idx0 + t) / KT, tj = (idx0 + t) % KT;          // (compile-time after unrolling)
                acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[s][ti], l[s][tj], acc0[t], 0, 0, 0);
                accr[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l[s][ti], l[s][tj], accr[t], 0, 0, 0);
                acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[s][ti], h[s][tj], acc0[t], 0, 0, 0);
                accr[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l[s][ti], h[s][tj], accr[t], 0, 0, 0);
            }
    }
#undef GFIX_REQUEST
    GFIX_STAMP(2);
    float* out = a.part + ((int64_t)f * GFIX_PARTS + blockIdx.x) * 2 * K * K;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int ti = (W * TPW + t) / KT, tj = (W * TPW + t) % KT;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kp = 32 * ti + (i & 3) + 8 * (i >> 2) + 4 * hi, k = 32 * tj + l31;
            out[kp * K + k] = acc0[t][i] * un;
            out[K * K + kp * K + k] = accr[t][i] * un;
        }
    }
    GFIX_STAMP(3);
}
// K = 128 (KT = 4): wave w owns tile row ti = w (its four tiles of both matrices) and needs ALL four 32-column chunks of every row.  Each wave
// loads and splits ONE chunk (its own) and the four exchange the fp16 fragments through LDS: a quarter of the loads and conversions of the
// scheme above, where every wave fetched and split whole rows (18.6 us at cfg4's share, most of it that).
template <int W>
__device__ __forceinline__ void gfix_gram_wave_k128(const GfixArgs& a, int f, float* red, gf16x8* frag) {
    constexpr int K = 128, KT = 4, NB = 8;   // steps of sixteen rows per batch: frag[step][chunk][term][lane], 64 KB
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int64_t rows = a.rows[f];
    const int64_t per = 16 * ((rows + 16 * GFIX_PARTS - 1) / (16 * GFIX_PARTS));
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float* X = a.X[f] + 32 * W + l31;
    f32x16 acc0[KT], accr[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[t][i] = 0.f; accr[t][i] = 0.f; }
    float v[NB][8];
#define GFIX_REQUEST(rb_)                                                                          \
    _Pragma("unroll") for (int s = 0; s < NB; ++s)                                                 \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                \
        const int64_t r = (rb_) + 16 * s + 8 * hi + q;                                             \
        v[s][q] = r < r1 ? X[r * K] : 0.f;                                                         \
    }
    if (r0 < r1) { GFIX_REQUEST(r0) }
    const float sc = gfix_scale(a.absmax, f, red), un = 1.f / (sc * sc);
    for (int64_t rb = r0; rb < r1; rb += 16 * NB) {
        gf16x8 h[NB], l[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = v[s][q] * sc;
                const _Float16 t = (_Float16)x;
                h[s][q] = t;
                l[s][q] = (_Float16)(x - (float)t);
            }
            frag[((s * KT + W) * 2 + 0) * 64 + lane] = h[s];
            frag[((s * KT + W) * 2 + 1) * 64 + lane] = l[s];
        }
        if (rb + 16 * NB < r1) { GFIX_REQUEST(rb + 16 * NB) }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NB; ++s)
#pragma unroll
            for (int tj = 0; tj < KT; ++tj) {
                const gf16x8 hj = tj == W ? h[s] : frag[((s * KT + tj) * 2 + 0) * 64 + lane];
                const gf16x8 lj = tj == W ? l[s] : frag[((s * KT + tj) * 2 + 1) * 64 + lane];
                acc0[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[s], lj, acc0[tj], 0, 0, 0);
                accr[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l[s], lj, accr[tj], 0, 0, 0);
                acc0[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[s], hj, acc0[tj], 0, 0, 0);
                accr[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(l[s], hj, accr[tj], 0, 0, 0);
            }
        __syncthreads();                     // (the fragments are rewritten by the next batch)
    }
#undef GFIX_REQUEST
    float* out = a.part + ((int64_t)f * GFIX_PARTS + blockIdx.x) * 2 * K * K;
#pragma unroll
    for (int tj = 0; tj < KT; ++tj)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kp = 32 * W + (i & 3) + 8 * (i >> 2) + 4 * hi, k = 32 * tj + l31;
            out[kp * K + k] = acc0[tj][i] * un;
            out[K * K + kp * K + k] = accr[tj][i] * un;
        }
}
template <int KT>
__global__ __launch_bounds__(256) void k_gfix_gram(GfixArgs a) {
    __shared__ float red[4];
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    if (!a.want[1 - f]) return;              // factor f's matrices correct the OTHER block's gradient
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (KT == 4) {
        extern __shared__ gf16x8 gfrag[];    // [8 steps][4 chunks][2 terms][64 lanes]: 64 KB
        if (w == 0) gfix_gram_wave_k128<0>(a, f, red, gfrag);
        else if (w == 1) gfix_gram_wave_k128<1>(a, f, red, gfrag);
        else if (w == 2) gfix_gram_wave_k128<2>(a, f, red, gfrag);
        else gfix_gram_wave_k128<3>(a, f, red, gfrag);
    } else {
        if (w == 0) gfix_gram_wave<KT, 0>(a, f, red);
        else if (w == 1) gfix_gram_wave<KT, 1>(a, f, red);
        else if (w == 2) gfix_gram_wave<KT, 2>(a, f, red);
        else gfix_gram_wave<KT, 3>(a, f, red);
    }
}

// entry e of Q[f][m]: four threads fold 32 partials each (all loads in flight), then the four sums in a fixed order
__global__ __launch_bounds__(256) void k_gfix_reduce(GfixArgs a) {
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    if (!a.want[1 - f]) return;
    GFIX_STAMP(4);
    const int n = 2 * a.K * a.K;
    const int t = blockIdx.x * 256 + threadIdx.x, e = t >> 2, q = t & 3;
    static_assert(GFIX_PARTS == 128, "four threads x 32 partials");
    double s = 0.0;
    if (e < n) {
        const float* p = a.part + (int64_t)f * GFIX_PARTS * n + (int64_t)(q * 32) * n + e;
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = p[(int64_t)i * n];
#pragma unroll
        for (int i = 0; i < 32; ++i) s += (double)v[i];
    }
    const int base = threadIdx.x & 60;
    double tot = __shfl(s, base);
#pragma unroll
    for (int i = 1; i < 4; ++i) tot += __shfl(s, base + i);
    if (e < n && q == 0) a.Q[(int64_t)f * n + e] = (float)tot;
    GFIX_STAMP(5);
}

// C_X = X Qr(Z) + x_r Q0(Z): one wave per (32 rows, 32 output columns).  Split-fp16 MFMA like K1's own contractions (exact-fp32 MFMA runs at
// the vector rate: the same product took 13 us at 16384 rows x 64, MFMA-bound): the rows as three fp16 terms of X 2^e (xh + xl + xm: xh is K1's
// a0 / s0, xl + xm = x_r to 2^-22), each matrix as two terms scaled by a power of two from its own maximum, transposed in LDS so that a lane's
// eight contraction indices are one 16-byte read;  X Qr ~ xh qh + xh ql + xl qh,  x_r Q0 ~ xl qh + xl ql + xm qh,  two accumulators (two scales).
template <int KT>
__global__ __launch_bounds__(KT == 4 ? 512 : 256) void k_gfix_apply(GfixArgs a) {
    constexpr int K = 32 * KT, LDQ = K + 8;  // halves per row of a transposed plane (+ 16 bytes: the 16-byte reads of 32 rows spread over the banks)
    constexpr int NT = KT == 4 ? 512 : 256, NW = NT / 64;      // K = 128: eight waves share one staging of the (4 x larger) matrices
    extern __shared__ _Float16 qpl[];        // [matrix: Qr, Q0][term][column k][k']
    __shared__ float red[8];
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;                // the block whose gradient is corrected
    if (!a.want[f]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int64_t rows = a.rows[f];
    const float* X = a.X[f];
    float* out = a.out[f];
    const int64_t ntask = (rows + 31) / 32 * KT, stride = (int64_t)gridDim.x * NW;     // a wave per task and round (launch_gfix: <= 256 workgroups, the matrices staged once each)
    if ((int64_t)blockIdx.x * NW >= ntask) return;
    GFIX_STAMP(6);
    f32x4 x[K / 16][2];                      // this lane's row, components 16 ks + 8 hi .. + 7 (the MFMA's A operand: i = lane & 31, k = 8 (lane >> 5) ..)
#define GFIX_LOAD_X(t_)                                                                                                   \
    {                                                                                                                     \
        const int64_t row_ = ((t_) / KT) * 32 + l31;                                                                      \
        const f32x4* xr = reinterpret_cast<const f32x4*>(X + (row_ < rows ? row_ : 0) * K + 8 * hi);                       \
        _Pragma("unroll") for (int ks = 0; ks < K / 16; ++ks) { x[ks][0] = xr[4 * ks]; x[ks][1] = xr[4 * ks + 1]; }       \
    }
    const int64_t task0 = (int64_t)blockIdx.x * NW + w;
    if (task0 < ntask) GFIX_LOAD_X(task0)
    // The matrices, read so that the TRANSPOSED planes are written with one 16-byte store per eight k': thread t takes column k = t % K and the
    // blocks of eight k' = 8 (t / K + (256 / K) i) .. -- 8 scalar loads per block, each coalesced across the lanes (consecutive k), one
    // conflict-free ds_write_b128 per term.  (A float4-per-thread read with sixty-four 2-byte scattered stores, 16-way bank conflicts, was the
    // kernel: 13 us of its 14.)
    constexpr int TPK = NT / K, NBLK = K / 8 / TPK;        // threads per column, blocks of eight k' per thread
    const int kcol = tid % K, kb0 = tid / K;
    float tq[2][NBLK][8];
    {
        const float* src = a.Q + (int64_t)(1 - f) * 2 * K * K;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < NBLK; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) tq[m][i][e] = src[(size_t)m * K * K + (size_t)(8 * (kb0 + TPK * i) + e) * K + kcol];
    }
    float mx[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < NBLK; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[m] = fmaxf(mx[m], fabsf(tq[m][i][e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx[0] = fmaxf(mx[0], __shfl_xor(mx[0], o)); mx[1] = fmaxf(mx[1], __shfl_xor(mx[1], o)); }
    __shared__ float redq[2][NW];
    if (lane == 0) { redq[0][w] = mx[0]; redq[1][w] = mx[1]; }
    const float sc = gfix_scale(a.absmax, f, red);        // (its barrier publishes redq as well)
    GFIX_STAMP(7);
    float sq[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float v = redq[m][0];
#pragma unroll
        for (int i = 1; i < NW; ++i) v = fmaxf(v, redq[m][i]);
        int q = 0;
        (void)frexpf(v, &q);
        sq[m] = ldexpf(1.f, v > 0.f ? 14 - q : 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {            // matrix m (0 = Q0, 1 = Qr) -> planes 2 (1 - m), + 1: Qr first
        _Float16* ph = qpl + (size_t)((1 - m) * 2) * K * LDQ + (size_t)kcol * LDQ;
        _Float16* pl = ph + (size_t)K * LDQ;
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            gf16x8 h8, l8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = tq[m][i][e] * sq[m];
                const _Float16 h = (_Float16)v;
                h8[e] = h;
                l8[e] = (_Float16)(v - (float)h);
            }
            *reinterpret_cast<gf16x8*>(ph + 8 * (kb0 + TPK * i)) = h8;
            *reinterpret_cast<gf16x8*>(pl + 8 * (kb0 + TPK * i)) = l8;
        }
    }
    __syncthreads();
    GFIX_STAMP(8);
    for (int64_t task = (int64_t)blockIdx.x * NW + w; task < ntask; task += stride) {
        const int64_t rt = task / KT;
        const int c = (int)(task % KT);
        const bool live = rt * 32 + l31 < rows;
        if (task != task0) GFIX_LOAD_X(task)   // (the first task's rows were requested before the matrices: one round trip less)
        f32x16 acc1, acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[i] = 0.f; acc2[i] = 0.f; }
        const _Float16* qb = qpl + (size_t)(32 * c + l31) * LDQ + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            gf16x8 xh, xl, xm;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (live ? x[ks][e >> 2][e & 3] : 0.f) * sc;
                const _Float16 h = (_Float16)v;
                const float r1 = v - (float)h;
                const _Float16 l = (_Float16)r1;
                xh[e] = h; xl[e] = l; xm[e] = (_Float16)(r1 - (float)l);
            }
            const gf16x8 qrh = *reinterpret_cast<const gf16x8*>(qb + 16 * ks);
            const gf16x8 qrl = *reinterpret_cast<const gf16x8*>(qb + (size_t)K * LDQ + 16 * ks);
            const gf16x8 q0h = *reinterpret_cast<const gf16x8*>(qb + (size_t)2 * K * LDQ + 16 * ks);
            const gf16x8 q0l = *reinterpret_cast<const gf16x8*>(qb + (size_t)3 * K * LDQ + 16 * ks);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, qrh, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xm, q0h, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, qrl, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, q0l, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, qrh, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, q0h, acc2, 0, 0, 0);
        }
        const float u1 = 1.f / (sc * sq[1]), u2 = 1.f / (sc * sq[0]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            if (r < rows) out[r * a.ld + 32 * c + l31] = acc1[i] * u1 + acc2[i] * u2;
        }
        GFIX_STAMP(9);
    }
#undef GFIX_LOAD_X
    GFIX_STAMP(10);
}

static hipError_t launch_gfix(const GfixArgs& a, hipStream_t s) {
    const int n = 2 * a.K * a.K;
    const size_t lds = (size_t)4 * a.K * (a.K + 8) * sizeof(_Float16);      // k_gfix_apply: two matrices x two fp16 terms, transposed
    const int64_t rmax = a.rows[0] > a.rows[1] ? a.rows[0] : a.rows[1];
    const int nw = a.K == 128 ? 8 : 4;                                               // waves per workgroup of k_gfix_apply
    unsigned nb = (unsigned)(((rmax + 31) / 32 * (a.K / 32) + nw - 1) / nw);         // k_gfix_apply: a wave per (32 rows, 32 columns) and round
    // one staging of the matrices per workgroup: at most as many workgroups as can be resident (K = 128: 139 KB of LDS = one per CU, two blocks: 128 each)
    if (nb > (a.K == 128 ? 128u : 256u)) nb = a.K == 128 ? 128 : 256;
    if (a.K == 64) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gfix_apply<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_gfix_gram<2>, dim3(GFIX_PARTS, 2), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_gfix_reduce, dim3((n * 4 + 255) / 256, 2), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_gfix_apply<2>, dim3(nb, 2), dim3(256), lds, s, a);
    } else if (a.K == 128) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gfix_apply<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_gfix_gram<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_gfix_gram<4>, dim3(GFIX_PARTS, 2), dim3(256), 64 * 1024, s, a);
        hipLaunchKernelGGL(k_gfix_reduce, dim3((n * 4 + 255) / 256, 2), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_gfix_apply<4>, dim3(nb, 2), dim3(512), lds, s, a);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
