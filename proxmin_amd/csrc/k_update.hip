// Factor-sized kernels of the nmf() hot path: everything that is O((M+N)K) per iteration.
//
//   - gradient-slab folding, PGM / FISTA update           (proxmin/algorithms.py:93-108,130-135)
//   - adaprox moment schemes, update, proximal sub-iterations (algorithms.py:147-245, :369-410)
//   - block-SDMM primal/dual update and residual norms     (proxmin/utils.py:295-391)
//   - the proximal operators themselves                    (proxmin/operators.py:20-160)
//
// Both blocks are tall row-major matrices with K components per row (A: M x K, St = S^T: N x K), so
// one set of kernels serves both.  A row is handled by LPR = 32 lanes (half a wavefront), lane l owning
// components l, l+32, l+64, l+96; the row sums inside prox_unity / prox_unity_plus are wavefront
// shuffles (5 xor steps), never LDS or atomics.  Reductions over the whole factor (norms for the
// stopping tests, max Psi, column sums) are written as one double per workgroup and folded by the next
// kernel in the chain in a fixed order, so results are deterministic.
//
// Control flow lives on the device: every kernel first reads DevStatus::halt (set by a single-block
// "decide" kernel when the run converged or needs more proximal sub-iterations than were enqueued)
// and becomes a no-op once it is set; the host enqueues chains of iterations without synchronising.
#include "pmx_common.h"

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// Cross-lane sums use DPP moves (ALU latency) where the hardware has the pattern and one LDS-crossbar swizzle /
// permute where it does not: five dependent ds_bpermute (what __shfl_xor compiles to) per row sum made the
// proximal sub-iteration passes latency-bound.  All lanes of the group end up with bit-identical totals.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;   // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int SWZ_XOR16 = (0x10 << 10) | 0x1f;                                                  // ds_swizzle bit mode

// Row sums keep the butterfly order 16, 8, 4, 2, 1 (the association the parity fixtures were pinned with): the
// three long strides as swizzles, the two short ones as DPP quad permutes.
constexpr int SWZ_XOR8 = (0x08 << 10) | 0x1f, SWZ_XOR4 = (0x04 << 10) | 0x1f;
__device__ __forceinline__ float row_sum32(float v) {   // sum over the 32 lanes that share a row
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), SWZ_XOR16));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), SWZ_XOR8));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), SWZ_XOR4));
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_XOR1>(v);
    return v;
}

__device__ __forceinline__ double swz16_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, SWZ_XOR16);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), SWZ_XOR16);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HALF_MIRROR>(v);
    v += dpp_d<DPP_MIRROR>(v);
    v += swz16_d(v);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_d<DPP_XOR1>(v));
    v = fmax(v, dpp_d<DPP_XOR2>(v));
    v = fmax(v, dpp_d<DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_d<DPP_MIRROR>(v));
    v = fmax(v, swz16_d(v));
    v = fmax(v, __shfl_xor(v, 32));
    return v;
}

template <int PAT>
__device__ __forceinline__ double swz_d(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)u, PAT);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(u >> 32), PAT);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// [r4] SIXTEEN wave sums at the price of about three: instead of 16 butterflies over all 64 lanes (16 x 6 adds, 16 x 8 DPP moves,
// 16 x 4 LDS-crossbar moves per lane -- 5 us for the 16 waves of a workgroup, measured in k_ada_tail), every exchange step HALVES
// the values a lane carries: with its partner at distance 1, 2, 4, 8 a lane keeps the values whose index bit equals its own lane bit
// and hands over the others (8 + 4 + 2 + 1 adds), then lane l holds the sum of value l & 15 over its 16 lanes and two butterfly
// steps finish it.  Every add has the same two operands as the butterfly's (x + y and y + x are the same bits): the result is
// wave_sum()'s, bit for bit.  Returns the total of value (lane & 15).
__device__ __forceinline__ double wave_sum16(const double (&v)[16]) {
    const int lane = threadIdx.x & 63;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    double w[8], x[4], y[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        w[i] = keep + dpp_d<DPP_XOR1>(send);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = b1 ? w[2 * i + 1] : w[2 * i], send = b1 ? w[2 * i] : w[2 * i + 1];
        x[i] = keep + dpp_d<DPP_XOR2>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b2 ? x[2 * i + 1] : x[2 * i], send = b2 ? x[2 * i] : x[2 * i + 1];
        y[i] = keep + swz_d<SWZ_XOR4>(send);
    }
    double z = (b3 ? y[1] : y[0]) + swz_d<SWZ_XOR8>(b3 ? y[0] : y[1]);
    z += swz16_d(z);
    z += __shfl_xor(z, 32);
    return z;
}

// block-wide sum of NV doubles; thread 0 writes dst[i * stride]
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&v)[NV], double* dst, int64_t stride, double* scratch /* >= NV*EW_WAVES */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * EW_WAVES + w] = v[i];
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int q = 0; q < EW_WAVES; ++q) s += scratch[threadIdx.x * EW_WAVES + q];
        dst[threadIdx.x * stride] = s;
    }
}

// fold one slot's EW_BLOCKS partials: every WAVE does it on its own (EW_BLOCKS/64 coalesced loads per
// lane + shuffles, no LDS, no barrier), in a fixed order, so every thread of the grid gets the same value.
__device__ __forceinline__ double fold_partials(const double* part, double* /*scratch*/) {
    const int lane = threadIdx.x & 63;
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < EW_BLOCKS / 64; ++i) v += part[i * 64 + lane];
    return wave_sum(v);
}
__device__ __forceinline__ double fold_partials_max(const double* part, double* /*scratch*/) {
    const int lane = threadIdx.x & 63;
    double v = -1.0;
#pragma unroll
    for (int i = 0; i < EW_BLOCKS / 64; ++i) v = fmax(v, part[i * 64 + lane]);
    return wave_max(v);
}

__device__ __forceinline__ double* part_ptr(double* partials, int slot, int blk) {
    return partials + ((int64_t)slot * 2 + blk) * EW_BLOCKS;
}

// Write-through (`sc1`, agent-scope relaxed) stores and loads of per-workgroup records: what one workgroup publishes for
// another INSIDE a launch (the last-arrival stopping test of k_pgm_update, the grid barriers of k_ada_tail) without
// release / acquire fences -- MI355X_MICROARCH.md, "sc0 sc1 stores and loads both sides".
__device__ __forceinline__ double sc1_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sc1_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block_sum_store / colsum_store with write-through stores, fold_partials with write-through loads: same trees
template <int NV>
__device__ __forceinline__ void block_sum_store_wt(double (&v)[NV], double* dst, int64_t stride, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[i * EW_WAVES + w] = v[i];
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int q = 0; q < EW_WAVES; ++q) s += scratch[threadIdx.x * EW_WAVES + q];
        sc1_store(dst + threadIdx.x * stride, s);
    }
}
__device__ __forceinline__ double fold_partials_wt(const double* part) {
    const int lane = threadIdx.x & 63;
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < EW_BLOCKS / 64; ++i) v += sc1_load(part + i * 64 + lane);
    return wave_sum(v);
}
__device__ __forceinline__ double nanmax(double a, double b);
__device__ __forceinline__ double wave_nanmax(double v);
__device__ __forceinline__ double fold_partials_nanmax_wt(const double* part) {
    const int lane = threadIdx.x & 63;
    double v = -1.0;
#pragma unroll
    for (int i = 0; i < EW_BLOCKS / 64; ++i) v = nanmax(v, sc1_load(part + i * 64 + lane));
    return wave_nanmax(v);
}

// ------------------------------------------------------------------------------------------------
// proximal operators on one row held across 32 lanes          (proxmin/operators.py:20-160)
// v[c] is component l32 + 32c; ok[c] says whether that component exists (< K).
// sk[c] is the step the solver passes for that component (scalar steps: all equal).
// ------------------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void prox_one(float (&v)[NC], const bool (&ok)[NC], const pmx_prox& p, const float (&sk)[NC]) {
    switch (p.op) {
        case PMX_PROX_ID: break;
        case PMX_PROX_ZERO:
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = 0.f;
            break;
        case PMX_PROX_PLUS:                                     // X[X<0] = 0
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = v[c] < 0.f ? 0.f : v[c];
            break;
        case PMX_PROX_UNITY:
        case PMX_PROX_UNITY_PLUS: {                             // X / sum(X, axis)   (no zero guard)
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (p.op == PMX_PROX_UNITY_PLUS) v[c] = v[c] < 0.f ? 0.f : v[c];
                s += ok[c] ? v[c] : 0.f;
            }
            s = row_sum32(s);
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = v[c] / s;
            break;
        }
        default: {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float th = (float)p.thresh;                           // (fp32 arithmetic: the threshold as the fp32 value it always was)
                const float t = p.relative ? th * sk[c] : th;               // operators.py:4-14
                float x = v[c];
                switch (p.op) {
                    case PMX_PROX_MIN: x = (x - t < 0.f) ? t : x; break;             // operators.py:66-68
                    case PMX_PROX_MAX: x = (x - t > 0.f) ? t : x; break;             // operators.py:82-84
                    case PMX_PROX_HARD: x = (fabsf(x) < t) ? 0.f : x; break;         // operators.py:125-127
                    case PMX_PROX_HARD_PLUS: x = (fabsf(x) < t) ? 0.f : x; x = x < 0.f ? 0.f : x; break;
                    case PMX_PROX_SOFT:
                    case PMX_PROX_SOFT_PLUS: {                                       // sign(X) * plus(|X| - t)
                        float m = fabsf(x) - t;
                        m = m < 0.f ? 0.f : m;
                        const float sg = (float)(x > 0.f) - (float)(x < 0.f);
                        x = sg * m;
                        if (p.op == PMX_PROX_SOFT_PLUS) x = x < 0.f ? 0.f : x;
                        break;
                    }
                    default: break;
                }
                v[c] = x;
            }
        }
    }
}

template <int NC>
__device__ __forceinline__ void prox_row(float (&v)[NC], const bool (&ok)[NC], const ProxSeq& ps, const float (&sk)[NC]) {
    for (int r = 0; r < ps.repeat; ++r)
        for (int q = 0; q < ps.n; ++q) prox_one<NC>(v, ok, ps.seq[q], sk);
}

// row iteration: half-wave h of the grid handles rows h, h + H, h + 2H, ...
#define ROW_LOOP_BEGIN(rows)                                                                     \
    const int l32 = threadIdx.x & 31;                                                            \
    const int64_t hw_ = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5;                    \
    const int64_t nhw_ = ((int64_t)EW_BLOCKS * EW_THREADS) >> 5;                                   \
    for (int64_t r = hw_; r < (rows); r += nhw_) {
#define ROW_LOOP_END }

struct SlabRef {
    const float* base;   // [n][stride]: slab i's row r at base + i * stride + r * K
    int n;
    int64_t stride;      // floats between slabs: rows x K of K1's FRAME (the factor's rows, or more: zero-padded frame of a ragged shape)
    int ld;              // floats between rows of a slab: K, or K1's padded K (pmx_k1_frame: a K without a tuned kernel runs the next one's)
    const float* extra = nullptr;   // [r5] one more slab outside the array (same row pitch), added LAST: the correction of a high x high residual (k_gfix.hip)
};

template <int NC>
__device__ __forceinline__ void load_grad(float (&g)[NC], const bool (&ok)[NC], const SlabRef& s, int64_t rows, int K, int64_t r, int l32) {
    // fixed summation order (slab 0, 1, 2, ...), but up to 16 values' loads are issued before the first add so that the fold is
    // bandwidth- rather than latency-bound (a dependent round trip to memory is ~1.3 us: a small problem's 13 gSt slabs in
    // batches of 4 were 5 of the 8 us of its moment phase)
#pragma unroll
    for (int c = 0; c < NC; ++c) g[c] = 0.f;
    const int64_t stride = s.stride;
    const float* p = s.base + r * s.ld + l32;
    int i = 0;
    constexpr int UB = NC == 1 ? 16 : (NC == 2 ? 8 : 4);
    for (; i + UB <= s.n; i += UB) {
        float v[UB][NC];
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) v[u][c] = ok[c] ? __builtin_nontemporal_load(p + u * stride + c * 32) : 0.f;   // read once
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) g[c] += v[u][c];
        p += UB * stride;
    }
    for (; i + 4 <= s.n; i += 4) {
        float v[4][NC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) v[u][c] = ok[c] ? __builtin_nontemporal_load(p + u * stride + c * 32) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) g[c] += v[u][c];
        p += 4 * stride;
    }
    for (; i < s.n; ++i) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) g[c] += p[c * 32];
        p += stride;
    }
    if (s.extra != nullptr) {
        const float* q = s.extra + r * s.ld + l32;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) g[c] += q[c * 32];
    }
}

// ------------------------------------------------------------------------------------------------
// fold gradient slabs into G (pmx_grad; also used by tests)
// ------------------------------------------------------------------------------------------------
struct FoldArgs {
    SlabRef slab[2];
    float* G[2];
    int64_t rows[2];
    int K;
    const DevStatus* status;
};
// [r4] K1's operands when the context's K has no tuned kernel: the factors copied into arrays with the next tuned K as row pitch
// (columns K .. Kk - 1 and the frame's extra rows were zeroed when the arrays were created and are never written)
struct PadArgs {
    const float* src[2];
    float* dst[2];
    int64_t rows[2];
    int K, Kk;
    const DevStatus* status;
};
__global__ __launch_bounds__(256) void k_pad_factors(PadArgs a) {
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int64_t n = a.rows[j] * a.K;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / a.K;
        const int k = (int)(e - r * a.K);
        a.dst[j][r * a.Kk + k] = a.src[j][e];
    }
}
void launch_pad_factors(const PadArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_pad_factors, dim3(512, 2), dim3(256), 0, s, a); }

template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_fold(FoldArgs a) {
    if (a.status != nullptr && chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int64_t rows = a.rows[j];
    const int K = a.K;
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float g[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
        load_grad<NC>(g, ok, a.slab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) a.G[j][r * K + l32 + 32 * c] = g[c];
    ROW_LOOP_END
}

// ------------------------------------------------------------------------------------------------
// standalone prox (operators.* called on an array)
// ------------------------------------------------------------------------------------------------
struct ProxArgs {
    float* X;
    int64_t rows;
    int K;
    ProxSeq prox;
    float stepk[MAXK];
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_prox_apply(ProxArgs a) {
    const int K = a.K;
    ROW_LOOP_BEGIN(a.rows)
        bool ok[NC];
        float v[NC], sk[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int kk = l32 + 32 * c;
            ok[c] = kk < K;
            v[c] = ok[c] ? a.X[r * K + kk] : 0.f;
            sk[c] = ok[c] ? a.stepk[kk] : 0.f;
        }
        prox_row<NC>(v, ok, a.prox, sk);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) a.X[r * K + l32 + 32 * c] = v[c];
    ROW_LOOP_END
}

// ------------------------------------------------------------------------------------------------
// PGM / FISTA update                                      (proxmin/algorithms.py:93-108,130-133)
//   X_new = prox(Xe - s G, s);  Xe_next = X_new + omega_next (X_new - X_old)
// ------------------------------------------------------------------------------------------------
struct PgmArgs {
    float* X[2];        // current iterate (updated in place)
    float* Xe[2];       // extrapolated point (== X when not accelerated)
    float* G[2];        // gradient output (returned by pgm, algorithms.py:144)
    SlabRef slab[2];
    int64_t rows[2];
    int K;
    ProxSeq prox[2];
    DevStatus* status;
    double* partials;
    int accelerated;
    float omega_next;
    // host round trip for a user-defined prox of block j (algorithms.py:107-108 with a Python callable):
    //   mode 0  the whole update in one launch (operators of this library);
    //   mode 1  "pre":  T_j = Xe_j - s_j G_j and nothing else -- the host applies the callable to T_j;
    //   mode 2  "post": the update with prox(...) := T_j as the host left it;      mode 3: block not touched by this launch
    int mode[2];
    float* T[2];
    // a user `step` that returned ARRAYS (algorithms.py:106-108: `_X[j] - S[j] * G[j]`, `prox[j](.., S[j])` broadcast like NumPy):
    // block j's step of every element, rows x K like the block itself (the host broadcasts what the callable returned); nullptr: the
    // scalar of DevStatus::step
    const float* stepArr[2];
    // stopping test by the LAST workgroup to finish (algorithms.py:130-135) instead of a separate single-workgroup launch:
    // every workgroup publishes its partial sums (release), takes a ticket, and the one that draws the last ticket of
    // this launch folds all partials in the usual fixed order and decides.  tickets == nullptr: the caller launches
    // k_pgm_decide itself (row-sharded runs, split iterations).
    unsigned* tickets;
    unsigned ticket_last;    // value the counter shows to the last arrival: tickets drawn by all launches so far, this one included, - 1
    float* absmax_out;       // as FinishArgs ([2][EW_BLOCKS] max |point the next K1 is evaluated at|), or nullptr; full grid only
    int nbx;                 // workgroups per block to launch (0: EW_BLOCKS).  Row r belongs to half-wave r mod 8192 of the full
                             // grid, so factors of <= 8192 rows need only ceil(rows / 32) workgroups; the others would
                             // contribute exact zeros to the partial sums (their entries already hold zeros) and one ticket
                             // each -- 512 atomics on one address are 5 us of a small problem's 13 us update
    double e_rel[2];
    // [r4] the NEXT iteration's step rule starts here: every row of the point the next gradient is evaluated at passes through
    // this kernel's registers, so workgroup b leaves the partial Gram matrix of ITS 32 rows -- k_gram_partial's own share and
    // summation order (gram_per: bit-identical partials) -- in gramPart[j][b]; k_gram_reduce folds the gram_nparts(rows) slots
    // that exist.  Only where one batch of rows per workgroup covers the factor (rows <= 4096 in both blocks) and K <= 64;
    // nullptr: off (k_gram_partial runs as before).
    float* gramPart;         // [2][GRAM_BLOCKS][KP*KP]
    int KP;
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_pgm_update(PgmArgs a) {
    __shared__ double scratch[2 * EW_WAVES];
    __shared__ int s_last;
    __shared__ float gtile[NC <= 2 ? 32 : 1][NC <= 2 ? 32 * NC + 1 : 1];   // the workgroup's 32 rows of the next evaluation point
    const int j = blockIdx.y;
    // (the halt flag and the step size travel together: one round trip to memory instead of two)
    const int halted = __builtin_nontemporal_load(&a.status->halt);
    const float s = (float)a.status->step[j];
    if (halted) return;
    if (a.mode[j] == 3) return;
    const int mode = a.mode[j];
    const int64_t rows = a.rows[j];
    const int K = a.K;
    const ProxSeq& px = a.prox[j];
    float* X = a.X[j];
    float* Xe = a.Xe[j];
    float d2 = 0.f, n2 = 0.f, xmax = 0.f;
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float g[NC], xo[NC], v[NC], sk[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
        load_grad<NC>(g, ok, a.slab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            xo[c] = ok[c] ? X[e] : 0.f;
            const float xe = a.accelerated ? (ok[c] ? Xe[e] : 0.f) : xo[c];
            const float se = a.stepArr[j] != nullptr ? (ok[c] ? a.stepArr[j][e] : 0.f) : s;
            v[c] = mode == 2 ? (ok[c] ? a.T[j][e] : 0.f) : xe - se * g[c];
            sk[c] = se;
        }
        if (mode == 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (ok[c]) a.T[j][r * K + l32 + 32 * c] = v[c];
            continue;
        }
        if (mode == 0) prox_row<NC>(v, ok, px, sk);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (ok[c]) {
                const int64_t e = r * K + l32 + 32 * c;
                X[e] = v[c];
                a.G[j][e] = g[c];
                float xnext = v[c];          // the point the next gradient is evaluated at
                if (a.accelerated) { xnext = v[c] + a.omega_next * (v[c] - xo[c]); Xe[e] = xnext; }
                if constexpr (NC <= 2) { if (a.gramPart != nullptr) gtile[threadIdx.x >> 5][l32 + 32 * c] = xnext; }
                xmax = fmaxf(xmax, fabsf(xnext));
                const float d = v[c] - xo[c];
                d2 += d * d;
                n2 += v[c] * v[c];
            }
        }
    ROW_LOOP_END
    if (mode == 1) return;
    if constexpr (NC <= 2) {
        if (a.gramPart != nullptr) {         // (uniform) partial Gram matrix of this workgroup's rows 32 b .. 32 b + 31
            const int KP = a.KP;
            const int64_t row0 = (int64_t)blockIdx.x * (EW_THREADS / 32);
            const int nrow = rows - row0 < EW_THREADS / 32 ? (int)(rows - row0) : EW_THREADS / 32;
            __syncthreads();
            float* out = a.gramPart + ((int64_t)j * GRAM_BLOCKS + blockIdx.x) * KP * KP;
            for (int e = threadIdx.x; e < KP * KP; e += EW_THREADS) {
                const int gi = e / KP, gj = e - gi * KP;
                float acc = 0.f;
                if (gi < K && gj < K)
                    for (int rr = 0; rr < nrow; ++rr) acc += gtile[rr][gi] * gtile[rr][gj];     // k_gram_partial's order: rows ascending
                out[e] = acc;          // (the fold reads gram_nparts(rows) = this grid's workgroups-with-rows slots)
            }
            __syncthreads();
        }
    }
    if (a.absmax_out != nullptr) {
        const double m = wave_max((double)xmax);
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double mm = scratch[0];
            for (int q = 1; q < EW_WAVES; ++q) mm = fmax(mm, scratch[q]);
            a.absmax_out[j * EW_BLOCKS + blockIdx.x] = (float)mm;
            for (int b = blockIdx.x + gridDim.x; b < EW_BLOCKS; b += gridDim.x) a.absmax_out[j * EW_BLOCKS + b] = 0.f;   // (a grid of nbx < 256)
        }
        __syncthreads();
    }
    double red[2] = {(double)d2, (double)n2};
    // SL_DIFF2 and SL_NORM2 are adjacent slots: stride between them = 2 * EW_BLOCKS doubles
    if (a.tickets == nullptr) {
        block_sum_store<2>(red, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
        return;
    }
    // the partial sums go out write-through, the ticket follows them (vmcnt(0): they have landed), and the last arrival
    // reads everybody's partials write-through as well: no fence (a release here would write back the whole L2 -- the
    // factors this kernel has just stored -- once per workgroup: 30 us at 4096 x 4096 x 32 instead of 12)
    block_sum_store_wt<2>(red, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == a.ticket_last;
    }
    __syncthreads();
    if (s_last) pgm_decide_body(a.status, a.partials, a.e_rel, 1, true);
}

// ------------------------------------------------------------------------------------------------
// PGM with backtracking line search (Beck & Teboulle eq. 3.2; proxmin/algorithms.py:110-127).
// k_bt_update (re)applies  X_j = prox(Xe_j - T_j s_j G_j, T_j s_j)  for the selected blocks, keeps X_ (the
// iterate before the step) in Xp, and reduces what the sufficient-decrease test needs:
//     sum (X - X_).G,  sum (X - X_)^2,  max|G|,  max|X_|   (+ sum X^2 for the stopping test).
// The first call of an iteration folds the gradient slabs into G; halvings re-read G.
// k_bt_collect folds the partials into DevStatus::bt for the host, which owns the while loop (each trial
// needs a fresh pass over Y for f(X), so a host round trip per trial is noise).
// ------------------------------------------------------------------------------------------------
struct BtArgs {
    float* X[2];
    const float* Xe[2];
    float* Xp[2];
    float* G[2];
    SlabRef slab[2];
    int64_t rows[2];
    int K;
    ProxSeq prox[2];
    DevStatus* status;
    double* partials;
    float T[2];
    int do_block[2];
    int first;           // first application in this iteration: X still holds X_, fold slabs, store Xp and G
    // [r4] a user-defined prox of block j inside the line search (one host round trip per TRIAL of that block):
    //   mode 0  the operator of this library, on the device;
    //   mode 1  "pre":  Tb_j = Xe_j - T_j s_j G_j for the host's callable (first: Xp and G stored as in mode 0); X_j untouched, no sums;
    //   mode 2  "post": X_j <- Tb_j as the host left it, and the sums of the trial (reads Xp and G: `first` must be 0)
    int mode[2];
    float* Tb[2];
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_bt_update(BtArgs a) {
    __shared__ double scratch[3 * EW_WAVES];
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    if (!a.do_block[j]) return;
    const int64_t rows = a.rows[j];
    const int K = a.K;
    const float s = a.T[j] * (float)a.status->step[j];      // T[j] * S[j]   (algorithms.py:108,125)
    float d1 = 0.f, d2 = 0.f, n2 = 0.f, mg = 0.f, mx = 0.f;
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float g[NC], xo[NC], v[NC], sk[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
        if (a.first) load_grad<NC>(g, ok, a.slab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            if (!a.first) g[c] = ok[c] ? a.G[j][e] : 0.f;
            xo[c] = ok[c] ? (a.first ? a.X[j][e] : a.Xp[j][e]) : 0.f;
            const float xe = ok[c] ? a.Xe[j][e] : 0.f;
            v[c] = a.mode[j] == 2 ? (ok[c] ? a.Tb[j][e] : 0.f) : xe - s * g[c];
            sk[c] = s;
        }
        if (a.mode[j] == 1) {                // the argument of the user's prox; X_ and G kept for the trials that follow
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (ok[c]) {
                    const int64_t e = r * K + l32 + 32 * c;
                    a.Tb[j][e] = v[c];
                    if (a.first) { a.Xp[j][e] = xo[c]; a.G[j][e] = g[c]; }
                }
            continue;
        }
        if (a.mode[j] == 0) prox_row<NC>(v, ok, a.prox[j], sk);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) {
                const int64_t e = r * K + l32 + 32 * c;
                a.X[j][e] = v[c];
                if (a.first) { a.Xp[j][e] = xo[c]; a.G[j][e] = g[c]; }
                const float d = v[c] - xo[c];
                d1 += d * g[c];
                d2 += d * d;
                n2 += v[c] * v[c];
                mg = fmaxf(mg, fabsf(g[c]));
                mx = fmaxf(mx, fabsf(xo[c]));
            }
    ROW_LOOP_END
    if (a.mode[j] == 1) return;              // (uniform per block: nothing to reduce yet)
    double red[2] = {(double)d2, (double)n2};
    block_sum_store<2>(red, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    double r1[1] = {(double)d1};
    block_sum_store<1>(r1, part_ptr(a.partials, SL_BT0, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    double m0 = wave_max((double)mg), m1 = wave_max((double)mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { scratch[threadIdx.x >> 6] = m0; scratch[EW_WAVES + (threadIdx.x >> 6)] = m1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int q = 0; q < EW_WAVES; ++q) { a0 = fmax(a0, scratch[q]); a1 = fmax(a1, scratch[EW_WAVES + q]); }
        part_ptr(a.partials, SL_BT0 + 1, j)[blockIdx.x] = a0;
        part_ptr(a.partials, SL_BT0 + 2, j)[blockIdx.x] = a1;
    }
}
struct BtCollectArgs {
    DevStatus* status;
    double* partials;
    int do_block[2];
};
__global__ __launch_bounds__(EW_THREADS) void k_bt_collect(BtCollectArgs a) {
    __shared__ double scratch[EW_WAVES];
    if (chain_halted(a.status)) return;
    for (int j = 0; j < 2; ++j) {
        if (!a.do_block[j]) continue;
        const double d1 = fold_partials(part_ptr(a.partials, SL_BT0, j), scratch);
        const double d2 = fold_partials(part_ptr(a.partials, SL_DIFF2, j), scratch);
        const double n2 = fold_partials(part_ptr(a.partials, SL_NORM2, j), scratch);
        const double mg = fold_partials_max(part_ptr(a.partials, SL_BT0 + 1, j), scratch);
        const double mx = fold_partials_max(part_ptr(a.partials, SL_BT0 + 2, j), scratch);
        if (threadIdx.x == 0) {
            a.status->bt[j][0] = d1; a.status->bt[j][1] = d2; a.status->bt[j][2] = mg; a.status->bt[j][3] = mx; a.status->bt[j][4] = n2;
        }
    }
}
// after the line search settled: FISTA extrapolation for the next iteration (algorithms.py:95) or plain copy
struct BtFinishArgs {
    const float* X[2];
    const float* Xp[2];
    float* Xe[2];
    int64_t rows[2];
    int K;
    const DevStatus* status;
    float omega_next;
};
__global__ __launch_bounds__(EW_THREADS) void k_bt_finish(BtFinishArgs a) {
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int64_t n = a.rows[j] * a.K;
    for (int64_t e = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; e < n; e += (int64_t)EW_BLOCKS * EW_THREADS) {
        const float x = a.X[j][e];
        a.Xe[j][e] = x + a.omega_next * (x - a.Xp[j][e]);
    }
}

// ------------------------------------------------------------------------------------------------
// Barzilai-Borwein step rule (utils.BarzilaiBorweinStepper, proxmin/utils.py:209-241), evaluated like the
// reference at the point the gradient was taken (the extrapolated point when accelerated).
//   k_bb_reduce: folds the gradient slabs into G (so the update kernel reads one array), forms
//                s = X - X_prev, y = G - G_prev, stores the new X_prev / G_prev, and reduces
//                sum s^2, sum s.y, sum y^2, sum G^2, max|X|, max|G| per workgroup;
//   k_bb_step  : it == 0 -> r * max|X| / max|G|; else min(|BB1 or BB2|, Delta / |G|) with Delta the
//                running minimum of |s| over it <= 3 (Burdakov et al. stabilisation).
// ------------------------------------------------------------------------------------------------
struct BBArgs {
    const float* X[2];      // point of evaluation
    SlabRef slab[2];
    float* G[2];            // folded gradient (output)
    float* Xprev[2];
    float* Gprev[2];
    int64_t rows[2];
    int K;
    const DevStatus* status;
    double* partials;
    int first;              // it == 0: nothing to difference against
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_bb_reduce(BBArgs a) {
    __shared__ double scratch[6 * EW_WAVES];
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int64_t rows = a.rows[j];
    const int K = a.K;
    float ss = 0.f, sy = 0.f, yy = 0.f, gg = 0.f, mx = 0.f, mg = 0.f;
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float g[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
        load_grad<NC>(g, ok, a.slab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) {
                const int64_t e = r * K + l32 + 32 * c;
                const float x = a.X[j][e];
                if (!a.first) {
                    const float sv = x - a.Xprev[j][e], yv = g[c] - a.Gprev[j][e];
                    ss += sv * sv;
                    sy += sv * yv;
                    yy += yv * yv;
                }
                gg += g[c] * g[c];
                mx = fmaxf(mx, fabsf(x));
                mg = fmaxf(mg, fabsf(g[c]));
                a.Xprev[j][e] = x;
                a.Gprev[j][e] = g[c];
                a.G[j][e] = g[c];
            }
    ROW_LOOP_END
    double red[4] = {(double)ss, (double)sy, (double)yy, (double)gg};
    block_sum_store<4>(red, part_ptr(a.partials, SL_BB0, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    double m0 = wave_max((double)mx), m1 = wave_max((double)mg);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { scratch[threadIdx.x >> 6] = m0; scratch[EW_WAVES + (threadIdx.x >> 6)] = m1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int q = 0; q < EW_WAVES; ++q) { a0 = fmax(a0, scratch[q]); a1 = fmax(a1, scratch[EW_WAVES + q]); }
        part_ptr(a.partials, SL_BB0 + 4, j)[blockIdx.x] = a0;
        part_ptr(a.partials, SL_BB0 + 5, j)[blockIdx.x] = a1;
    }
}
struct BBStepArgs {
    DevStatus* status;
    double* partials;
    int it, type;
    double init_r;
};
__global__ __launch_bounds__(EW_THREADS) void k_bb_step(BBStepArgs a) {
    __shared__ double scratch[EW_WAVES];
    if (chain_halted(a.status)) return;
    for (int j = 0; j < 2; ++j) {
        const double ss = fold_partials(part_ptr(a.partials, SL_BB0 + 0, j), scratch);
        const double sy = fold_partials(part_ptr(a.partials, SL_BB0 + 1, j), scratch);
        const double yy = fold_partials(part_ptr(a.partials, SL_BB0 + 2, j), scratch);
        const double gg = fold_partials(part_ptr(a.partials, SL_BB0 + 3, j), scratch);
        const double mx = fold_partials_max(part_ptr(a.partials, SL_BB0 + 4, j), scratch);
        const double mg = fold_partials_max(part_ptr(a.partials, SL_BB0 + 5, j), scratch);
        if (threadIdx.x == 0) {
            DevStatus* st = a.status;
            double step;
            if (a.it == 0) {
                st->bb_delta[j] = 1e300 * 1e300;              // inf (utils.py:219)
                step = a.init_r * mx / mg;                    // utils.py:222
            } else {
                const double bb = a.type == 1 ? ss / sy : sy / yy;              // utils.py:231-234
                if (a.it <= 3) st->bb_delta[j] = fmin(st->bb_delta[j], sqrt(ss));   // utils.py:237-238
                step = fmin(fabs(bb), st->bb_delta[j] / sqrt(gg));              // utils.py:239-241
            }
            st->step[j] = step;
        }
    }
}

// single-block decision kernel shared by pgm and adaprox outer tests (algorithms.py:130-135,403-410)
// ---- stand-alone Barzilai-Borwein sums of ONE block on caller arrays (utils.BarzilaiBorweinStepper.step called as a function,
// proxmin/utils.py:216-241): s = X - X_prev, y = G - G_prev in the arrays' own type, products and sums in fp64, fixed order ----------
constexpr int BBS_BLOCKS = 256;
// np.max propagates NaN (utils.py:222: a diverged run gets a NaN step in the reference); fmax() drops it
__device__ __forceinline__ double bbs_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : fmax(a, b); }
template <typename T>
__global__ __launch_bounds__(256) void k_bb_sums(const T* X, const T* Xp, const T* G, const T* Gp, int64_t n, double* part) {
    __shared__ double red[4][6];
    double s2 = 0.0, sy = 0.0, y2 = 0.0, g2 = 0.0, mx = 0.0, mg = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)BBS_BLOCKS * 256) {
        const T x = X[i], g = G[i];
        const T s = Xp ? (T)(x - Xp[i]) : (T)0, y = Gp ? (T)(g - Gp[i]) : (T)0;
        s2 += (double)s * (double)s;
        sy += (double)s * (double)y;
        y2 += (double)y * (double)y;
        g2 += (double)g * (double)g;
        mx = bbs_max(mx, fabs((double)x));
        mg = bbs_max(mg, fabs((double)g));
    }
    double v[6] = {s2, sy, y2, g2, mx, mg};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v[q], o); v[q] = q < 4 ? v[q] + t : bbs_max(v[q], t); }
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 6; ++q) red[threadIdx.x >> 6][q] = v[q];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int q = threadIdx.x;
        double r = red[0][q];
        for (int w = 1; w < 4; ++w) r = q < 4 ? r + red[w][q] : bbs_max(r, red[w][q]);
        part[blockIdx.x * 6 + q] = r;
    }
}
__global__ void k_bb_sums_fold(const double* part, double* out) {
    const int q = threadIdx.x;
    if (q >= 6) return;
    double r = part[q];
    for (int b = 1; b < BBS_BLOCKS; ++b) r = q < 4 ? r + part[b * 6 + q] : bbs_max(r, part[b * 6 + q]);
    out[q] = r;
}

struct DecideArgs {
    DevStatus* status;
    double* partials;
    double e_rel[2];
    int check;          // evaluate the convergence test
};
__device__ __forceinline__ void pgm_decide_body(DevStatus* st, double* partials, const double (&e_rel)[2], int check, bool wt) {
    double d[2], n[2];
    {   // the four sums' partials are requested together (one memory latency), then folded in fold_partials' order
        const int lane = threadIdx.x & 63;
        const double* q[4] = {part_ptr(partials, SL_DIFF2, 0), part_ptr(partials, SL_NORM2, 0), part_ptr(partials, SL_DIFF2, 1), part_ptr(partials, SL_NORM2, 1)};
        double u[4][EW_BLOCKS / 64];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < EW_BLOCKS / 64; ++i) u[k][i] = wt ? sc1_load(q[k] + i * 64 + lane) : q[k][i * 64 + lane];
        double r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < EW_BLOCKS / 64; ++i) v += u[k][i];
            r[k] = wave_sum(v);
        }
        d[0] = r[0]; n[0] = r[1]; d[1] = r[2]; n[1] = r[3];
    }
    if (threadIdx.x == 0) {
        int all = 1;
        for (int j = 0; j < 2; ++j) {
            const int c = d[j] <= e_rel[j] * e_rel[j] * n[j];
            st->conv[j] = c;
            st->norms[j][0] = d[j];
            st->norms[j][1] = n[j];
            all &= c;
        }
        st->it_done += 1;
        if (check && all) {
            st->stopped = 1;
            st->reason = HALT_CONVERGED;
            __threadfence();
            st->halt = 1;
        }
    }
}
__global__ __launch_bounds__(EW_THREADS) void k_pgm_decide(DecideArgs a) {
    if (chain_halted(a.status)) return;
    pgm_decide_body(a.status, a.partials, a.e_rel, a.check, false);
}

// ------------------------------------------------------------------------------------------------
// column sums (adaprox step rule, nmf.py:91-93) -- partial per workgroup
// ------------------------------------------------------------------------------------------------
struct ColsumArgs {
    const float* X[2];
    int64_t rows[2];
    int K;
    double* colpart;      // [2][EW_BLOCKS][MAXK]
    const DevStatus* status;
};
template <int NC>
__device__ __forceinline__ void colsum_store(const float (&cs)[NC], double* colpart, int j, float* sm /* [8][MAXK] */) {
    const int hwi = threadIdx.x >> 5, l32 = threadIdx.x & 31;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) sm[hwi * MAXK + l32 + 32 * c] = cs[c];
    __syncthreads();
    if (threadIdx.x < 32 * NC) {
        double s = 0.0;
        for (int h = 0; h < EW_THREADS / 32; ++h) s += (double)sm[h * MAXK + threadIdx.x];
        colpart[((int64_t)j * EW_BLOCKS + blockIdx.x) * MAXK + threadIdx.x] = s;
    }
}
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_colsum(ColsumArgs a) {
    __shared__ float sm[(EW_THREADS / 32) * MAXK];
    if (a.status != nullptr && chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int K = a.K;
    float cs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cs[c] = 0.f;
    ROW_LOOP_BEGIN(a.rows[j])
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (l32 + 32 * c < K) cs[c] += a.X[j][r * K + l32 + 32 * c];
    ROW_LOOP_END
    colsum_store<NC>(cs, a.colpart, j, sm);
}

// alpha_k = mean_k / 10 from the column-sum partials (single block)
struct AlphaArgs {
    DevStatus* status;
    const double* colpart;
    int64_t rows_global[2];   // divisor of the mean (global M for a row-sharded A)
    int K;
    int use_fixed;
    float fixed[2];
    const float* comm_colsum;  // row-sharded runs: all-reduced column sums of A (K floats), else nullptr
};
// Column sums of block j from the per-workgroup partials, by the whole workgroup (EW_THREADS = 8 groups x MAXK
// components): thread (k, grp) adds partials grp, grp + 8, ... in that order, then the 8 group sums are added in
// order.  All of a thread's loads are issued before the first add: the partials were written by other XCDs'
// workgroups, every access is a trip to memory, and a load-add-load-add loop made this the longest part of the
// single-workgroup kernels (k_ada_decide: 18 us).  Result in asum[0..NG)[k]; ends with a barrier.
constexpr int ALPHA_NG = EW_THREADS / MAXK;
__device__ __forceinline__ void colsum_fold(const double* colpart_j, int K, bool skip, double (*asum)[MAXK]) {
    const int t = threadIdx.x, k = t & (MAXK - 1), grp = t >> 7;
    constexpr int PER = EW_BLOCKS / ALPHA_NG;
    double v[PER];
    const bool on = k < K && !skip;
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = on ? colpart_j[(int64_t)(grp + ALPHA_NG * i) * MAXK + k] : 0.0;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    __syncthreads();
    asum[grp][k] = s;
    __syncthreads();
}
__device__ __forceinline__ void compute_alpha(const AlphaArgs& a) {
    __shared__ double asum[ALPHA_NG][MAXK];
    const int t = threadIdx.x;
    if (a.use_fixed == 2) return;            // a user `step` callable: the host writes DevStatus::alpha before every iteration
    for (int j = 0; j < 2; ++j) {
        colsum_fold(a.colpart + (int64_t)j * EW_BLOCKS * MAXK, a.K, a.use_fixed || (j == 0 && a.comm_colsum != nullptr), asum);
        if (t < a.K) {
            float al;
            if (a.use_fixed) al = a.fixed[j];
            else {
                double tot = 0.0;
                if (j == 0 && a.comm_colsum != nullptr) tot = (double)a.comm_colsum[t];
                else
                    for (int q = 0; q < ALPHA_NG; ++q) tot += asum[q][t];
                al = (float)(tot / (double)a.rows_global[j]) / 10.f;
            }
            a.status->alpha[j][t] = al;
        }
    }
}
__global__ __launch_bounds__(EW_THREADS) void k_alpha_init(AlphaArgs a) {
    if (chain_halted(a.status)) return;
    compute_alpha(a);
}

// prox_unity / prox_unity_plus ALONG THE ROWS (numpy axis = 0 of a rows x K array: every component's column is divided
// by its sum, operators.py:41-52), for the stand-alone operator entry points: k_colsum's per-workgroup partial column sums
// (block 0 of `colpart`), folded here by every workgroup in colsum_fold's fixed order, then the scaling.  No zero guard,
// like the reference.  `plus`: the projection onto the non-negative numbers comes first (it has been applied by the caller
// before the column sums were taken); this kernel only divides.
struct ColScaleArgs {
    float* X;
    int64_t rows;
    int K;
    const double* colpart;   // [EW_BLOCKS][MAXK]
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_colscale(ColScaleArgs a) {
    __shared__ double asum[ALPHA_NG][MAXK];
    __shared__ float tot[MAXK];
    colsum_fold(a.colpart, a.K, false, asum);
    if (threadIdx.x < MAXK) {
        double t = 0.0;
        for (int q = 0; q < ALPHA_NG; ++q) t += asum[q][threadIdx.x];
        tot[threadIdx.x] = (float)t;
    }
    __syncthreads();
    const int K = a.K;
    ROW_LOOP_BEGIN(a.rows)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int kk = l32 + 32 * c;
            if (kk < K) a.X[r * K + kk] = a.X[r * K + kk] / tot[kk];
        }
    ROW_LOOP_END
}

// ------------------------------------------------------------------------------------------------
// adaprox: moments + update                      (proxmin/algorithms.py:147-245, :369-378)
// ------------------------------------------------------------------------------------------------
struct MomentArgs {
    float* X[2];
    float* Xp[2];       // copy of the pre-update iterate (only when check_convergence)
    float* Mm[2];
    float* Vv[2];
    float* Vh[2];       // nullptr unless warm-started (algorithms.py:356-359)
    float* Psi[2];      // written when the block has a prox (sub-iterations need it)
    SlabRef slab[2];
    int64_t rows[2];
    int K;
    DevStatus* status;
    double* partials;
    int scheme;
    int it;
    double b1t, b1prev, b2, eps, p;
    int check_convergence;
    int has_prox[2];
};
// scalars of the moment schemes for iteration `it` (algorithms.py:147-245), uniform over the grid
struct MomentScalars {
    double b1, bias1, rho;
    float c1, c2, bias2, epsf, rfac, adamx_factor;
};
__device__ __forceinline__ MomentScalars moment_scalars(const MomentArgs& a) {
    MomentScalars s;
    s.b1 = a.b1t;
    s.c1 = (float)(1.0 - a.b2);                                   // python floats are weak scalars: fp32 arithmetic
    s.c2 = (float)a.b2;
    const double t = (double)(a.it + 1);
    s.bias1 = 1.0 - pow(s.b1, t);                                  // 1 - b1[it]**t
    s.bias2 = (float)(1.0 - pow(a.b2, t));                         // 1 - b2**t
    s.epsf = (float)a.eps;
    // radam scalars (algorithms.py:224-245)
    const double rho_inf = 2.0 / (1.0 - a.b2) - 1.0;
    s.rho = rho_inf - 2.0 * t * pow(a.b2, t) / (1.0 - pow(a.b2, t));
    s.rfac = s.rho > 4.0 ? (float)sqrt((s.rho - 4.0) * (s.rho - 2.0) * rho_inf / (rho_inf - 4.0) / (rho_inf - 2.0) / s.rho) : 1.f;
    s.adamx_factor = (float)(((1.0 - s.b1) * (1.0 - s.b1)) / ((1.0 - a.b1prev) * (1.0 - a.b1prev)));
    return s;
}
// One element of block j: gradient gg, step alpha; updates M, V (, Vhat) in memory and returns the updated iterate
// x - alpha Phi / Psi (algorithms.py:375-378) and Psi.
__device__ __forceinline__ float moment_elem(const MomentArgs& a, const MomentScalars& s, int j, int64_t e, float gg, float alpha, float xo, float& psi_out) {
    float* Mm = a.Mm[j];
    float* Vv = a.Vv[j];
    float* Vh = a.Vh[j];
    // b1[it] is a NumPy float64 scalar in the reference, so M is formed in fp64 and rounded on store
    const float m = (float)((1.0 - s.b1) * (double)gg + s.b1 * (double)Mm[e]);
    const float v = s.c1 * (gg * gg) + s.c2 * Vv[e];
    Mm[e] = m;
    Vv[e] = v;
    double upd;   // alpha * Phi / Psi
    float psi;
    switch (a.scheme) {
        case PMX_ADAM:
            psi = sqrtf(v / s.bias2) + s.epsf;
            upd = (double)alpha * ((double)m / s.bias1) / (double)psi;
            break;
        case PMX_NADAM:
            psi = sqrtf(v / s.bias2) + s.epsf;
            upd = (double)alpha * ((s.b1 * (double)m + (1.0 - s.b1) * (double)gg) / s.bias1) / (double)psi;
            break;
        case PMX_RADAM:
            psi = s.rho > 4.0 ? sqrtf(v / s.bias2) / s.rfac : 1.f;
            if (s.epsf > 0.f) psi = fmaxf(psi, sqrtf(s.epsf));
            upd = (double)alpha * ((double)m / s.bias1) / (double)psi;
            break;
        default: {   // amsgrad / padam / adamx (algorithms.py:170-221)
            float cap = v;
            if (Vh != nullptr) {
                const float old = Vh[e];
                cap = fmaxf(a.scheme == PMX_ADAMX ? s.adamx_factor * old : old, v);
                Vh[e] = cap;
            }
            if (s.epsf > 0.f) cap = fmaxf(cap, s.epsf);
            psi = a.scheme == PMX_PADAM ? powf(cap, (float)a.p) : sqrtf(cap);
            upd = (double)(alpha * m / psi);
        }
    }
    psi_out = psi;
    return (float)((double)xo - upd);
}
// max that lets a NaN through, like np.max (fmaxf / fmax drop NaNs): Psi goes NaN with the iterate, and the reference's
// gamma = Alpha / np.max(Psi) then poisons the whole proximal loop visibly instead of continuing on a partly NaN block
__device__ __forceinline__ float nanmaxf(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : fmax(a, b); }
__device__ __forceinline__ double wave_nanmax(double v) {
    v = nanmax(v, dpp_d<DPP_XOR1>(v));
    v = nanmax(v, dpp_d<DPP_XOR2>(v));
    v = nanmax(v, dpp_d<DPP_HALF_MIRROR>(v));
    v = nanmax(v, dpp_d<DPP_MIRROR>(v));
    v = nanmax(v, swz16_d(v));
    v = nanmax(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ double fold_partials_nanmax(const double* part) {
    const int lane = threadIdx.x & 63;
    double v = -1.0;
#pragma unroll
    for (int i = 0; i < EW_BLOCKS / 64; ++i) v = nanmax(v, part[i * 64 + lane]);
    return wave_nanmax(v);
}

template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_ada_moment(MomentArgs a) {
    __shared__ double scratch[EW_WAVES];
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    const int64_t rows = a.rows[j];
    const int K = a.K;
    float* X = a.X[j];
    const MomentScalars ms = moment_scalars(a);
    float alpha[NC];
    {
        const int l32_ = threadIdx.x & 31;
#pragma unroll
        for (int c = 0; c < NC; ++c) alpha[c] = (l32_ + 32 * c < K) ? a.status->alpha[j][l32_ + 32 * c] : 0.f;
    }
    float maxpsi = -1.f;
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float g[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
        load_grad<NC>(g, ok, a.slab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!ok[c]) continue;
            const int64_t e = r * K + l32 + 32 * c;
            const float xo = X[e];
            float psi;
            const float xn = moment_elem(a, ms, j, e, g[c], alpha[c], xo, psi);
            if (a.check_convergence) a.Xp[j][e] = xo;
            X[e] = xn;
            if (a.has_prox[j]) a.Psi[j][e] = psi;
            maxpsi = nanmaxf(maxpsi, psi);
        }
    ROW_LOOP_END
    double mv = (double)maxpsi;
    mv = wave_nanmax(mv);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mv;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = scratch[0];
        for (int q = 1; q < EW_WAVES; ++q) m = nanmax(m, scratch[q]);
        part_ptr(a.partials, SL_MAXPSI, j)[blockIdx.x] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// adaprox: one proximal sub-iteration pass                 (proxmin/algorithms.py:383-400)
//   z_new = prox(z - gamma/alpha * Psi * (z - X), gamma);  stop when |z_new - z|^2 <= e^2 |z|^2
// Pass t (0-based) is launched as its own kernel; before doing any work it evaluates the stopping
// test of pass t-1 from that pass's per-workgroup partial sums (all workgroups fold them in the same
// order and reach the same verdict; workgroup 0 publishes it for the host and for later kernels).
// ------------------------------------------------------------------------------------------------
struct SubArgs {
    float* X[2];
    float* Psi[2];
    float* zb[2][2];
    int64_t rows[2];
    int K;
    ProxSeq prox[2];
    DevStatus* status;
    double* partials;
    double e_rel[2];
    int t;               // index of the first pass of this launch (finish: number of passes enqueued)
    int nt;              // passes per launch (1 or SUB_NT_MAX), the same for every launch of an iteration
    int prox_max_iter;
    int has_prox[2];
};

// If the loop of block j finished BEFORE pass a.t, returns the number of passes it took (tau >= 1); otherwise 0.
// Uniform across the workgroup (and across workgroups).  The passes to judge are those of the previous launch,
// [a.t - a.nt, a.t): wave w folds the sum (pass w >> 1, kind w & 1) in the usual fixed order, then every thread
// scans the passes in order.  Earlier launches were judged by their successors (verdict published in DevStatus).
__device__ __forceinline__ int sub_finished_before(const SubArgs& a, int j, double* scratch /* >= 2 * SUB_NT_MAX */) {
    if (a.t == 0) return 0;
    __shared__ int pub;
    if (threadIdx.x == 0) pub = a.status->sub_done[j] ? a.status->sub_tau[j] : 0;   // published by an earlier kernel
    __syncthreads();
    const int tau_pub = pub;
    __syncthreads();   // `pub` is rewritten by the next call (k_ada_finish judges both blocks): every wave must have read it
    if (tau_pub > 0) return tau_pub;
    const int t0 = a.t - a.nt;
    const int w = threadIdx.x >> 6;
    if (w < 2 * a.nt) {
        const int pass = t0 + (w >> 1);
        const double v = fold_partials(part_ptr(a.partials, SL_SUBR0 + 2 * (pass % SUB_RING) + (w & 1), j), scratch);
        if ((threadIdx.x & 63) == 0) scratch[w] = v;
    }
    __syncthreads();
    int tau = 0;
    for (int q = 0; q < a.nt && tau == 0; ++q) {
        const double d = scratch[2 * q], n = scratch[2 * q + 1];
        if ((d <= a.e_rel[j] * a.e_rel[j] * n) || (t0 + q + 1 >= a.prox_max_iter)) tau = t0 + q + 1;
    }
    __syncthreads();   // scratch is reused by the caller
    if (tau > 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        a.status->sub_tau[j] = tau;
        __threadfence();
        a.status->sub_done[j] = 1;
    }
    return tau;
}

// step size data of the sub-iteration of block j (algorithms.py:384): gamma = Alpha / max(Psi), ratio = gamma / Alpha
template <int NC>
__device__ __forceinline__ void sub_steps(const SubArgs& a, int j, double* scratch, float (&gam)[NC], float (&rat)[NC]) {
    (void)scratch;
    const double maxpsi = fold_partials_nanmax(part_ptr(a.partials, SL_MAXPSI, j));
    const int l32_ = threadIdx.x & 31;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int kk = l32_ + 32 * c;
        const float al = kk < a.K ? a.status->alpha[j][kk] : 0.f;
        gam[c] = al / (float)maxpsi;
        rat[c] = gam[c] / al;                     // NaN if alpha == 0, as in the reference
    }
}

// NT passes per launch, z kept in registers between them (every supported prox acts within a row, and a row lives
// in one 32-lane group); per pass the two sums of the stopping test go to that pass's ring slot.  The launch writes
// only the state after its LAST pass: if the loop turns out to have ended inside the launch, k_ada_finish replays
// the few passes from the launch's (untouched) input.  Launch b reads X (b = 0) or zb[(b-1) & 1] and writes zb[b & 1].
template <int NC, int NT>
__global__ __launch_bounds__(EW_THREADS) void k_ada_sub(SubArgs a) {
    __shared__ double scratch[2 * NT * EW_WAVES];
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y;
    if (!a.has_prox[j]) return;
    if (sub_finished_before(a, j, scratch) > 0) return;
    const int64_t rows = a.rows[j];
    const int K = a.K;
    const int b = a.t / NT;
    const float* zc = b == 0 ? a.X[j] : a.zb[j][(b - 1) & 1];
    float* zn = a.zb[j][b & 1];
    float gam[NC], rat[NC];
    sub_steps<NC>(a, j, scratch, gam, rat);
    float d2[NT], n2[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { d2[q] = 0.f; n2[q] = 0.f; }
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float z[NC], x[NC], ps[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            ok[c] = l32 + 32 * c < K;
            z[c] = ok[c] ? zc[e] : 0.f;
            x[c] = ok[c] ? a.X[j][e] : 0.f;
            ps[c] = ok[c] ? a.Psi[j][e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            float v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = z[c] - rat[c] * ps[c] * (z[c] - x[c]);
            prox_row<NC>(v, ok, a.prox[j], gam);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (ok[c]) {
                    const float d = v[c] - z[c];
                    d2[q] += d * d;
                    n2[q] += z[c] * z[c];
                }
                z[c] = v[c];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) zn[r * K + l32 + 32 * c] = z[c];
    ROW_LOOP_END
    double red[2 * NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { red[2 * q] = (double)d2[q]; red[2 * q + 1] = (double)n2[q]; }
    // a.t is a multiple of NT and SUB_RING of SUB_NT_MAX: the NT ring slots of this launch are consecutive
    block_sum_store<2 * NT>(red, part_ptr(a.partials, SL_SUBR0 + 2 * (a.t % SUB_RING), j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
}

// ------------------------------------------------------------------------------------------------
// adaprox: finish an iteration -- X <- z, outer norms, column sums   (algorithms.py:400-410)
// `t` = number of sub-iteration passes enqueued so far for this iteration.
// ------------------------------------------------------------------------------------------------
struct FinishArgs {
    SubArgs s;
    float* Xp[2];
    double* colpart;
    int check_convergence;
    float* absmax_out;   // [2][EW_BLOCKS] per-workgroup max |X_j| of the iterate written here (the fp16 K1's operand
                         // scales, saving its own k_absmax pass), or nullptr
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_ada_finish(FinishArgs a) {
    __shared__ double scratch[2 * EW_WAVES];
    __shared__ float sm[(EW_THREADS / 32) * MAXK];
    if (chain_halted(a.s.status)) return;
    const int j = blockIdx.y;
    const int64_t rows = a.s.rows[j];
    const int K = a.s.K;
    DevStatus* st = a.s.status;
    const float* src = a.s.X[j];
    int replay = 0;                      // passes to redo from `src` (the loop ended inside a launch)
    float gam[NC], rat[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { gam[c] = 0.f; rat[c] = 0.f; }
    // Nothing is written until BOTH blocks' loops have ended: a replay below starts from X itself, so finishing
    // one block now and coming back for the other (after the host fed it more passes) would apply it twice.
    int tau_of[2] = {0, 0};
    bool need_more = false;
    for (int jj = 0; jj < 2; ++jj) {
        if (!a.s.has_prox[jj]) continue;
        tau_of[jj] = sub_finished_before(a.s, jj, scratch);
        if (tau_of[jj] == 0) {
            // more passes are needed than were enqueued: leave everything untouched; k_ada_decide halts
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) st->need_sub[jj] = 1;
            need_more = true;
        }
    }
    if (need_more) return;
    if (a.s.has_prox[j]) {
        const int tau = tau_of[j];
        const int b = (tau - 1) / a.s.nt, q = (tau - 1) % a.s.nt;
        if (q == a.s.nt - 1) {
            src = a.s.zb[j][b & 1];                              // the launch's own output
        } else {
            src = b == 0 ? a.s.X[j] : a.s.zb[j][(b - 1) & 1];    // its input, q + 1 passes to redo
            replay = q + 1;
            sub_steps<NC>(a.s, j, scratch, gam, rat);
        }
    }
    float* X = a.s.X[j];
    float d2 = 0.f, n2 = 0.f, xmax = 0.f;
    float cs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cs[c] = 0.f;
    ROW_LOOP_BEGIN(rows)
        bool ok[NC];
        float z[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ok[c] = l32 + 32 * c < K;
            z[c] = ok[c] ? src[r * K + l32 + 32 * c] : 0.f;
        }
        if (replay > 0) {
            float x[NC], ps[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int64_t e = r * K + l32 + 32 * c;
                x[c] = ok[c] ? X[e] : 0.f;
                ps[c] = ok[c] ? a.s.Psi[j][e] : 0.f;
            }
            for (int q = 0; q < replay; ++q) {
                float v[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = z[c] - rat[c] * ps[c] * (z[c] - x[c]);
                prox_row<NC>(v, ok, a.s.prox[j], gam);
#pragma unroll
                for (int c = 0; c < NC; ++c) z[c] = v[c];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (ok[c]) {
                const int64_t e = r * K + l32 + 32 * c;
                const float x = z[c];
                X[e] = x;
                xmax = fmaxf(xmax, fabsf(x));
                if (a.check_convergence) {
                    const float d = x - a.Xp[j][e];
                    d2 += d * d;
                    n2 += x * x;
                }
                cs[c] += x;
            }
        }
    ROW_LOOP_END
    double red[2] = {(double)d2, (double)n2};
    block_sum_store<2>(red, part_ptr(a.s.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    colsum_store<NC>(cs, a.colpart, j, sm);
    if (a.absmax_out != nullptr) {
        const double m = wave_max((double)xmax);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double mm = scratch[0];
            for (int q = 1; q < EW_WAVES; ++q) mm = fmax(mm, scratch[q]);
            a.absmax_out[j * EW_BLOCKS + blockIdx.x] = (float)mm;
        }
    }
}

// single-block end-of-iteration kernel for adaprox
struct AdaDecideArgs {
    AlphaArgs al;
    double* partials;
    double e_rel[2];
    int check_convergence;
    int has_prox[2];
    int host_tau[2];     // passes of a proximal loop that ran around a user-defined prox on the host (block j), else 0
};
__global__ __launch_bounds__(EW_THREADS) void k_ada_decide(AdaDecideArgs a) {
    __shared__ double scratch[EW_WAVES];
    DevStatus* st = a.al.status;
    if (chain_halted(st)) return;
    __shared__ int need;
    if (threadIdx.x == 0) need = st->need_sub[0] | st->need_sub[1];
    __syncthreads();
    if (need) {   // the enqueued proximal sub-iterations did not suffice: stop the chain, the host resumes
        if (threadIdx.x == 0) {
            st->reason = HALT_NEED_SUB;
            __threadfence();
            st->halt = 1;
        }
        return;
    }
    double d[2] = {0, 0}, n[2] = {0, 0};
    if (a.check_convergence)
        for (int j = 0; j < 2; ++j) {
            d[j] = fold_partials(part_ptr(a.partials, SL_DIFF2, j), scratch);
            n[j] = fold_partials(part_ptr(a.partials, SL_NORM2, j), scratch);
        }
    compute_alpha(a.al);   // step sizes for the NEXT iteration from the updated factors (nmf.py:93)
    if (threadIdx.x == 0) {
        int all = 1;
        for (int j = 0; j < 2; ++j) {
            const int c = a.check_convergence ? (d[j] <= a.e_rel[j] * a.e_rel[j] * n[j]) : 0;
            st->conv[j] = c;
            st->norms[j][0] = d[j];
            st->norms[j][1] = n[j];
            all &= c;
            const int tau = a.has_prox[j] ? st->sub_tau[j] : a.host_tau[j];
            st->sub_total[j] += tau;
            st->last_tau[j] = tau;
            st->sub_done[j] = 0;
            st->sub_tau[j] = 0;
        }
        st->it_done += 1;
        if (a.check_convergence && all) {
            st->stopped = 1;
            st->reason = HALT_CONVERGED;
            __threadfence();
            st->halt = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adaprox: the whole iteration tail as ONE persistent kernel                 (algorithms.py:374-410)
//
//   moment + update  ->  [B1]  ->  proximal sub-iterations, 4 passes per round  ->  [B2, one per round]  ->  X <- z,
//   outer norms, column sums  ->  [B3]  ->  next iteration's step sizes + end-of-iteration bookkeeping
//
// instead of the four launches k_ada_moment / k_ada_sub / k_ada_finish / k_ada_decide.  One workgroup per CU, the same
// rows per thread and the same reduction trees as those kernels: results are bit-identical to the chain of launches.
// What the fusion removes is memory traffic and dependent round trips, not arithmetic: a workgroup's rows of the
// updated iterate, of Psi and of the proximal iterate z never leave its LDS between the phases (three arrays, 4 bytes x
// 1024 threads per row slot), and the only data that crosses workgroups inside the launch are the per-workgroup partial
// sums the phases already exchanged -- a few hundred bytes per workgroup and barrier.  Those records are stored and
// loaded write-through (agent-scope relaxed atomics = `sc1`), so the grid barrier is a pure arrival count with NO release
// / acquire fences: hierarchical (8 group counters of gridDim.x / 8 arrivals -> one top counter -> one generation word
// per group, all monotonic over the life of the context, MI355X_MICROARCH.md "barrier-xcd"), placement-independent.
// All workgroups must be co-resident.  Nothing promises that (another process may hold CUs), so the launch opens with a
// census barrier B0 whose outcome is decided by ONE compare-and-swap: either the last arrival wins it (GO), or a
// workgroup that has waited 2 ms does (ABORT) -- in which case nothing has been written yet, every workgroup returns,
// DevStatus::tail_fault stops the chain and the host goes back to the separate kernels for good (pmx_api.hip).  B0's
// arrival is issued first thing and awaited after the launch's scalar set-up, before the first store.
// ------------------------------------------------------------------------------------------------
struct GridBar {                 // zeroed at context creation
    unsigned cnt[8][16];         // arrivals per group (64-byte spacing)
    unsigned top[16];
    unsigned gen[8][16];         // 2 x (barriers completed) [+ 1: aborted] as published to each group
    unsigned verdict[16];        // census: 2 e = GO, 2 e + 1 = ABORT (CAS from the previous barrier's value)
    unsigned epoch[16];          // barriers completed when the last launch that used the barrier ended
};
__device__ __forceinline__ unsigned gb_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gb_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one thread per workgroup.  e = number of this barrier (1-based over the context's life).
__device__ __forceinline__ void gb_arrive(GridBar* b, unsigned e, bool census) {
    const unsigned ng = 8, per = gridDim.x / ng, g = blockIdx.x & 7;
    const unsigned a = __hip_atomic_fetch_add(&b->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == per * e) {
        const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == ng * e) {
            unsigned word = 2 * e;
            if (census) {
                unsigned expect = 2 * (e - 1);
                if (!__hip_atomic_compare_exchange_strong(&b->verdict[0], &expect, word, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) word = expect;   // an impatient workgroup aborted first
            } else {
                gb_store(&b->verdict[0], word);
            }
            for (unsigned q = 0; q < ng; ++q) gb_store(&b->gen[q][0], word);
        }
    }
}
// returns the generation word seen (2 e: go, 2 e + 1: aborted); give_up_ticks > 0: after that many 100 MHz ticks try to abort
__device__ __forceinline__ unsigned gb_wait(GridBar* b, unsigned e, long long give_up_ticks) {
    const unsigned g = blockIdx.x & 7;
    const long long t0 = wall_clock64();
    for (;;) {
        const unsigned v = gb_load(&b->gen[g][0]);
        if ((v >> 1) >= e) return v;
        __builtin_amdgcn_s_sleep(4);
        if (give_up_ticks > 0 && wall_clock64() - t0 > give_up_ticks) {
            unsigned expect = 2 * (e - 1);
            if (__hip_atomic_compare_exchange_strong(&b->verdict[0], &expect, 2 * e + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                for (unsigned q = 0; q < 8; ++q) gb_store(&b->gen[q][0], 2 * e + 1);
                return 2 * e + 1;
            }
            give_up_ticks = 0;          // the last arrival got there first: the word is on its way
        }
    }
}
// full barrier for a phase boundary: every wave drains its own (write-through) stores, then one thread arrives and waits.
// After a passed census every workgroup is resident, so the wait is bounded only as a last line of defence (100 ms: a
// workgroup that died): the launch then marks the chain failed (tail_fault = 2, the host raises) instead of spinning on.
__device__ __forceinline__ void gb_sync(GridBar* b, unsigned e, DevStatus* st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        gb_arrive(b, e, false);
        const unsigned g = blockIdx.x & 7;
        const long long t0 = wall_clock64();
        for (unsigned spins = 1; (gb_load(&b->gen[g][0]) >> 1) < e; ++spins) {
            __builtin_amdgcn_s_sleep(4);
            if ((spins & 1023u) == 0 && wall_clock64() - t0 > 10000000) {
                st->tail_fault = 2;
                st->reason = HALT_ERROR;
                __threadfence();
                st->halt = 1;
                break;
            }
        }
    }
    __syncthreads();
}

constexpr int TAIL_NT = 4;       // proximal passes per round (the launch size the chain of kernels settles on)
struct TailArgs {
    MomentArgs m;                // moment phase (slabs, M, V, Vhat, X, Xp, scheme scalars); m.Psi is not used
    ProxSeq prox[2];
    double e_rel[2];
    int prox_max_iter;
    double* colpart;
    float* absmax_out;           // as FinishArgs
    AlphaArgs al;                // next iteration's step sizes (as AdaDecideArgs)
    int decide_check;            // evaluate the outer stopping test here (0 in row-sharded runs: k_shard_post does)
    GridBar* bar;
    int slots[2];                // LDS row slots per thread of block j: ceil(rows[j] / 8192)
    long long* prof;             // tuning (PMX_TAIL_PROF=1): 100 MHz time stamps of workgroup 0 at the phase boundaries, else nullptr
    int prof_fine;               // PMX_TAIL_PROF=2: three more stamps inside the proximal passes' phase (passes A | sums A | passes S | sums S)
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_ada_tail(TailArgs a) {
    __shared__ double scratch[4 * TAIL_NT * EW_WAVES];
    __shared__ double dbuf[EW_BLOCKS + 8];
    __shared__ unsigned s_word;
    extern __shared__ float lds[];           // [3 arrays: X, z, Psi][slots[0] + slots[1]][NC][EW_THREADS]
    static_assert(EW_BLOCKS == 2 * MAXK, "the step-size phase gives one (block, component) to each workgroup");
    DevStatus* st = a.m.status;
    if (chain_halted(st)) return;
    const int tid = threadIdx.x, l32 = tid & 31;
    const int K = a.m.K;
    GridBar* bar = a.bar;
    unsigned ep = bar->epoch[0];             // barriers completed before this launch (written by its predecessor's last act)
    int nstamp = 0;
    auto stamp = [&]() { if (a.prof != nullptr && blockIdx.x == 0 && tid == 0) a.prof[nstamp++] = wall_clock64(); };
    stamp();
    if (tid == 0) gb_arrive(bar, ep + 1, true);          // census B0: awaited before the first store
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + tid) >> 5;
    const int64_t nhw = ((int64_t)EW_BLOCKS * EW_THREADS) >> 5;
    const int nslot = a.slots[0] + a.slots[1];
    float* Lx = lds;                                       // updated iterate before the prox (the loop's fixed X)
    float* Lz = lds + (size_t)nslot * NC * EW_THREADS;     // proximal iterate at the start of the current round
    float* Lp = lds + (size_t)2 * nslot * NC * EW_THREADS; // Psi; last, at least 16 KB: the finish phase's column-sum scratch
    auto slot = [&](int j, int i, int c) { return ((size_t)((j ? a.slots[0] : 0) + i) * NC + c) * EW_THREADS + tid; };
    bool ok[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
    const MomentScalars ms = moment_scalars(a.m);
    const bool any_prox = a.m.has_prox[0] || a.m.has_prox[1];

    // ---------------- phase M: moments and update (k_ada_moment) -------------------------------------------------
    // (the first rows' loads are requested before the census wait, whose latency they hide)
    float g_pre[NC], x_pre[NC];
    const bool pre = a.slots[0] > 0 && hw < a.m.rows[0];
    if (pre) {
        load_grad<NC>(g_pre, ok, a.m.slab[0], a.m.rows[0], K, hw, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) x_pre[c] = ok[c] ? a.m.X[0][hw * K + l32 + 32 * c] : 0.f;
    }
    // what the previous launch (or the host) left in the control block, requested before the census wait as well: this iteration's
    // step sizes (nmf.py:93) and the pass counts of the last iteration ([r4] they used to be loaded where they are first needed --
    // a round trip to memory in front of the moment phase and another in front of the proximal passes)
    float alpha0[NC], alpha1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        alpha0[c] = ok[c] ? st->alpha[0][l32 + 32 * c] : 0.f;
        alpha1[c] = ok[c] ? st->alpha[1][l32 + 32 * c] : 0.f;
    }
    const int last_tau0 = st->last_tau[0], last_tau1 = st->last_tau[1];
    {   // census B0: every workgroup must be resident before the first store of the launch
        if (tid == 0) s_word = gb_wait(bar, ep + 1, 200000);
        __syncthreads();
        if (s_word & 1u) {
            if (blockIdx.x == 0 && tid == 0) {
                st->tail_fault = 1;
                st->reason = HALT_ERROR;
                __threadfence();
                st->halt = 1;
            }
            return;
        }
        ep += 1;
    }
    stamp();
    float mpsi0 = -1.f, mpsi1 = -1.f;
    for (int j = 0; j < 2; ++j) {
        const int64_t rows = a.m.rows[j];
        float alpha[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) alpha[c] = j ? alpha1[c] : alpha0[c];
        float maxpsi = -1.f;
        for (int i = 0; i < a.slots[j]; ++i) {
            const int64_t r = hw + (int64_t)i * nhw;
            if (r >= rows) break;
            float g[NC], xo[NC];
            if (j == 0 && i == 0 && pre) {
#pragma unroll
                for (int c = 0; c < NC; ++c) { g[c] = g_pre[c]; xo[c] = x_pre[c]; }
            } else {
                load_grad<NC>(g, ok, a.m.slab[j], rows, K, r, l32);
#pragma unroll
                for (int c = 0; c < NC; ++c) xo[c] = ok[c] ? a.m.X[j][r * K + l32 + 32 * c] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!ok[c]) continue;
                const int64_t e = r * K + l32 + 32 * c;
                float psi;
                const float xn = moment_elem(a.m, ms, j, e, g[c], alpha[c], xo[c], psi);
                if (a.m.check_convergence) a.m.Xp[j][e] = xo[c];
                Lx[slot(j, i, c)] = xn;
                Lp[slot(j, i, c)] = psi;
                maxpsi = nanmaxf(maxpsi, psi);
            }
        }
        if (j) mpsi1 = maxpsi; else mpsi0 = maxpsi;
    }
    {   // [r4] both blocks' maxima of Psi in one staged reduction (one barrier pair behind BOTH row loops, none between them)
        const double mv0 = wave_nanmax((double)mpsi0), mv1 = wave_nanmax((double)mpsi1);
        __syncthreads();
        if ((tid & 63) == 0) { scratch[tid >> 6] = mv0; scratch[EW_WAVES + (tid >> 6)] = mv1; }
        __syncthreads();
        if (tid == 0 || tid == 64) {
            const int jj = tid >> 6;
            double m = scratch[jj * EW_WAVES];
            for (int q = 1; q < EW_WAVES; ++q) m = nanmax(m, scratch[jj * EW_WAVES + q]);
            sc1_store(part_ptr(a.m.partials, SL_MAXPSI, jj) + blockIdx.x, m);
        }
    }

    // ---------------- phase S: proximal sub-iterations (k_ada_sub + the replay of k_ada_finish) -------------------
    int tau0 = 0, tau1 = 0;
    stamp();
    if (any_prox) {
        gb_sync(bar, ++ep, st);              // B1: every workgroup's max Psi
        stamp();
        bool done0 = !a.m.has_prox[0], done1 = !a.m.has_prox[1];
        // max Psi per block (algorithms.py:384), kept in two scalars; gamma and gamma / alpha are re-derived where needed
        double mp0, mp1;
        {   // both blocks' partials requested before the first of them is reduced: one memory latency, not two
            const int lane = tid & 63;
            const double* q0 = part_ptr(a.m.partials, SL_MAXPSI, 0);
            const double* q1 = part_ptr(a.m.partials, SL_MAXPSI, 1);
            double u0[EW_BLOCKS / 64], u1[EW_BLOCKS / 64];
#pragma unroll
            for (int i = 0; i < EW_BLOCKS / 64; ++i) { u0[i] = sc1_load(q0 + i * 64 + lane); u1[i] = sc1_load(q1 + i * 64 + lane); }
            double v0 = -1.0, v1 = -1.0;
#pragma unroll
            for (int i = 0; i < EW_BLOCKS / 64; ++i) { v0 = nanmax(v0, u0[i]); v1 = nanmax(v1, u1[i]); }
            mp0 = wave_nanmax(v0);
            mp1 = wave_nanmax(v1);
        }
        stamp();
        auto steps = [&](int j, float (&gam)[NC], float (&rat)[NC]) {
            const float mp = (float)(j ? mp1 : mp0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float al = j ? alpha1[c] : alpha0[c];
                gam[c] = al / mp;
                rat[c] = gam[c] / al;                    // NaN if alpha == 0, as in the reference
            }
        };
        // up to TAIL_NT passes (n of them) from the round's starting iterate; SUMS: also the two sums of the stopping test
        auto passes = [&](int j, int i, int n, bool first_round, const float (&gam)[NC], const float (&rat)[NC], float (&z)[NC],
                          auto sums_c, float (&d2)[TAIL_NT], float (&n2)[TAIL_NT]) {
            constexpr bool SUMS = decltype(sums_c)::value;
            float x[NC], ps[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                x[c] = Lx[slot(j, i, c)];
                ps[c] = Lp[slot(j, i, c)];
                z[c] = first_round ? x[c] : Lz[slot(j, i, c)];
                if (!ok[c]) { x[c] = 0.f; ps[c] = 0.f; z[c] = 0.f; }
            }
#pragma unroll
            for (int q = 0; q < TAIL_NT; ++q) {
                if (q < n) {
                    float v[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) v[c] = z[c] - rat[c] * ps[c] * (z[c] - x[c]);
                    prox_row<NC>(v, ok, a.prox[j], gam);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (SUMS && ok[c]) {
                            const float d = v[c] - z[c];
                            d2[q] += d * d;
                            n2[q] += z[c] * z[c];
                        }
                        z[c] = v[c];
                    }
                }
            }
        };
        using yes = std::integral_constant<bool, true>;
        using no = std::integral_constant<bool, false>;
        // Rounds of up to TAIL_NT passes per block, one barrier per round.  Round 0 runs as many passes as the block's loop
        // took in the previous iteration (the steady state repeats itself) and leaves its end state in Lz: if the loop
        // ends exactly there, nothing is recomputed.  It starts from Lx, which stays intact, so a loop that ends earlier
        // inside round 0 is replayed from there; later rounds (start-up transient only) run TAIL_NT passes for the sums
        // alone and then redo the passes that count from their starting state.  Pass numbers and sums do not depend on
        // how the passes are grouped into rounds: bit-identical to one launch per pass.
        int t0_0 = 0, t0_1 = 0;                   // passes done before the current round, per block
        for (int round = 0; !(done0 && done1); ++round) {
            int nt_0 = TAIL_NT, nt_1 = TAIL_NT;
            if (round == 0) {
                const int l0 = last_tau0, l1 = last_tau1;
                nt_0 = l0 >= 1 && l0 <= TAIL_NT ? l0 : TAIL_NT;
                nt_1 = l1 >= 1 && l1 <= TAIL_NT ? l1 : TAIL_NT;
            }
            // [r4] ONE workgroup-wide reduction for the sums of BOTH blocks (the passes of A, then of S, then 16 values through the
            // same tree each -- wave sum, 16-term serial fold: bit-identical partials): two reductions cost two barrier pairs and
            // twice the wait for the slowest wave, 2.2 us each in the phase stamps
            double red[2][2 * TAIL_NT];
#pragma unroll
            for (int q = 0; q < 2 * TAIL_NT; ++q) { red[0][q] = 0.0; red[1][q] = 0.0; }
            for (int j = 0; j < 2; ++j) {
                if (j ? done1 : done0) continue;
                const int nt = j ? nt_1 : nt_0;
                float gam[NC], rat[NC];
                steps(j, gam, rat);
                float d2[TAIL_NT], n2[TAIL_NT];
#pragma unroll
                for (int q = 0; q < TAIL_NT; ++q) { d2[q] = 0.f; n2[q] = 0.f; }
                for (int i = 0; i < a.slots[j]; ++i) {
                    if (hw + (int64_t)i * nhw >= a.m.rows[j]) break;
                    float z[NC];
                    passes(j, i, nt, round == 0, gam, rat, z, yes{}, d2, n2);
                    if (round == 0) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) Lz[slot(j, i, c)] = z[c];
                    }
                }
#pragma unroll
                for (int q = 0; q < TAIL_NT; ++q) {
                    if (j) { red[1][2 * q] = (double)d2[q]; red[1][2 * q + 1] = (double)n2[q]; }
                    else { red[0][2 * q] = (double)d2[q]; red[0][2 * q + 1] = (double)n2[q]; }
                }
                if (round == 0 && a.prof_fine) stamp();
            }
            {
                static_assert(4 * TAIL_NT == 16, "wave_sum16");
                const int lane = tid & 63, w = tid >> 6;
                double all[16];
#pragma unroll
                for (int q = 0; q < 2 * TAIL_NT; ++q) { all[q] = red[0][q]; all[2 * TAIL_NT + q] = red[1][q]; }
                const double tot = wave_sum16(all);          // value (lane & 15), wave_sum()'s bits
                __syncthreads();
                if (lane < 16) scratch[lane * EW_WAVES + w] = tot;
                __syncthreads();
                if (tid < 4 * TAIL_NT) {
                    const int jj = tid / (2 * TAIL_NT), q = tid - jj * 2 * TAIL_NT;
                    if (!(jj ? done1 : done0)) {
                        double sum = 0.0;
                        for (int u = 0; u < EW_WAVES; ++u) sum += scratch[tid * EW_WAVES + u];
                        sc1_store(part_ptr(a.m.partials, SL_SUBR0 + 2 * TAIL_NT * (round & 3), jj) + blockIdx.x + (int64_t)q * 2 * EW_BLOCKS, sum);
                    }
                }
            }
            if (round == 0) stamp();
            gb_sync(bar, ++ep, st);          // B2: the round's sums
            if (round == 0) stamp();
            for (int j = 0; j < 2; ++j) {
                if (j ? done1 : done0) continue;
                const int nt = j ? nt_1 : nt_0, t0 = j ? t0_1 : t0_0;
                const int w = tid >> 6;
                __syncthreads();
                if (w < 2 * nt) {
                    const double v = fold_partials_wt(part_ptr(a.m.partials, SL_SUBR0 + 2 * TAIL_NT * (round & 3) + w, j));
                    if ((tid & 63) == 0) scratch[w] = v;
                }
                __syncthreads();
                int tau = 0;
                for (int q = 0; q < nt && tau == 0; ++q) {
                    const double d = scratch[2 * q], n = scratch[2 * q + 1];
                    if ((d <= a.e_rel[j] * a.e_rel[j] * n) || (t0 + q + 1 >= a.prox_max_iter)) tau = t0 + q + 1;
                }
                __syncthreads();
                const int keep = tau > 0 ? tau - t0 : nt;            // passes of this round that count
                if (!(round == 0 && keep == nt)) {                   // (round 0 already left its end state in Lz)
                    float gam[NC], rat[NC], dd[TAIL_NT], nn[TAIL_NT];
                    steps(j, gam, rat);
                    for (int i = 0; i < a.slots[j]; ++i) {
                        if (hw + (int64_t)i * nhw >= a.m.rows[j]) break;
                        float z[NC];
                        passes(j, i, keep, round == 0, gam, rat, z, no{}, dd, nn);
#pragma unroll
                        for (int c = 0; c < NC; ++c) Lz[slot(j, i, c)] = z[c];
                    }
                }
                if (j) { t0_1 = t0 + nt; if (tau > 0) { done1 = true; tau1 = tau; } }
                else { t0_0 = t0 + nt; if (tau > 0) { done0 = true; tau0 = tau; } }
            }
        }
    }

    // ---------------- phase F: X <- z, outer norms, column sums (k_ada_finish) --------------------------------------
    stamp();
    float* sm = Lp;                      // colsum_store's scratch (Psi is no longer needed)
    // [r4] both blocks' rows first, then ONE staged reduction for everything this phase sums (outer norms, maxima, column sums): four
    // workgroup barriers instead of fourteen (each block used to run its own three reductions); every sum through its own tree as
    // before (wave tree + 16-term fold; column sums: 32 half-waves in order): bit-identical to k_ada_finish
    float f_d2[2] = {0.f, 0.f}, f_n2[2] = {0.f, 0.f}, f_xmax[2] = {0.f, 0.f};
    float cs0[NC], cs1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { cs0[c] = 0.f; cs1[c] = 0.f; }
    for (int j = 0; j < 2; ++j) {
        const int64_t rows = a.m.rows[j];
        const float* src = a.m.has_prox[j] ? Lz : Lx;
        float d2 = 0.f, n2 = 0.f, xmax = 0.f;
        float cs[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) cs[c] = 0.f;
        for (int i = 0; i < a.slots[j]; ++i) {
            const int64_t r = hw + (int64_t)i * nhw;
            if (r >= rows) break;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!ok[c]) continue;
                const int64_t e = r * K + l32 + 32 * c;
                const float x = src[slot(j, i, c)];
                a.m.X[j][e] = x;
                xmax = fmaxf(xmax, fabsf(x));
                if (a.m.check_convergence) {
                    const float d = x - a.m.Xp[j][e];
                    d2 += d * d;
                    n2 += x * x;
                }
                cs[c] += x;
            }
        }
        f_d2[j] = d2; f_n2[j] = n2; f_xmax[j] = xmax;
#pragma unroll
        for (int c = 0; c < NC; ++c) { if (j) cs1[c] = cs[c]; else cs0[c] = cs[c]; }
    }
    {
        const int lane = tid & 63, w = tid >> 6, hwi = tid >> 5;
        double r4[4] = {wave_sum((double)f_d2[0]), wave_sum((double)f_n2[0]), wave_sum((double)f_d2[1]), wave_sum((double)f_n2[1])};
        const double m0 = wave_max((double)f_xmax[0]), m1 = wave_max((double)f_xmax[1]);
        __syncthreads();                                     // (the rows' reads of Lp's neighbours are done: sm may be overwritten)
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) scratch[q * EW_WAVES + w] = r4[q];
            scratch[4 * EW_WAVES + w] = m0;
            scratch[5 * EW_WAVES + w] = m1;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) sm[hwi * MAXK + l32 + 32 * c] = cs0[c];
        __syncthreads();
        if (tid < 32 * NC) {                                  // column sums of block 0 (colsum_store's order)
            double t = 0.0;
            for (int h = 0; h < EW_THREADS / 32; ++h) t += (double)sm[h * MAXK + tid];
            sc1_store(a.colpart + ((int64_t)0 * EW_BLOCKS + blockIdx.x) * MAXK + tid, t);
        }
        if (tid >= 64 * 8 && tid < 64 * 8 + 4) {              // (another wave: the four outer sums; block_sum_store's order)
            const int q = tid - 64 * 8;
            double t = 0.0;
            for (int u = 0; u < EW_WAVES; ++u) t += scratch[q * EW_WAVES + u];
            sc1_store(part_ptr(a.m.partials, SL_DIFF2, q >> 1) + blockIdx.x + (int64_t)(q & 1) * 2 * EW_BLOCKS, t);
        }
        if (a.absmax_out != nullptr && tid >= 64 * 9 && tid < 64 * 9 + 2) {
            const int q = tid - 64 * 9;
            double mm = scratch[(4 + q) * EW_WAVES];
            for (int u = 1; u < EW_WAVES; ++u) mm = fmax(mm, scratch[(4 + q) * EW_WAVES + u]);
            a.absmax_out[q * EW_BLOCKS + blockIdx.x] = (float)mm;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) sm[hwi * MAXK + l32 + 32 * c] = cs1[c];
        __syncthreads();
        if (tid < 32 * NC) {                                  // column sums of block 1
            double t = 0.0;
            for (int h = 0; h < EW_THREADS / 32; ++h) t += (double)sm[h * MAXK + tid];
            sc1_store(a.colpart + ((int64_t)1 * EW_BLOCKS + blockIdx.x) * MAXK + tid, t);
        }
    }
    stamp();
    gb_sync(bar, ++ep, st);                  // B3: column sums and outer norms of every workgroup
    stamp();

    // ---------------- phase D: next step sizes, one (block, component) per workgroup; bookkeeping (k_ada_decide) ------
    {
        const int j = blockIdx.x / MAXK, k = blockIdx.x % MAXK;
        if (k < K && a.al.use_fixed != 2) {
            float al;
            if (a.al.use_fixed) al = a.al.fixed[j];
            else {
                double tot;
                if (j == 0 && a.al.comm_colsum != nullptr) tot = (double)a.al.comm_colsum[k];
                else {
                    // colsum_fold's order: 8 groups, group q adds the partials q, q + 8, ... in that order, then the groups in order
                    if (tid < EW_BLOCKS) dbuf[tid] = sc1_load(a.colpart + ((int64_t)j * EW_BLOCKS + tid) * MAXK + k);
                    __syncthreads();
                    if (tid < ALPHA_NG) {
                        double s = 0.0;
                        for (int i = 0; i < EW_BLOCKS / ALPHA_NG; ++i) s += dbuf[tid + ALPHA_NG * i];
                        dbuf[EW_BLOCKS + tid] = s;
                    }
                    __syncthreads();
                    tot = 0.0;
                    for (int q = 0; q < ALPHA_NG; ++q) tot += dbuf[EW_BLOCKS + q];
                }
                al = (float)(tot / (double)a.al.rows_global[j]) / 10.f;
            }
            if (tid == 0) st->alpha[j][k] = al;
        }
    }
    if (blockIdx.x == 0) {
        double d[2] = {0, 0}, n[2] = {0, 0};
        if (a.m.check_convergence)
            for (int j = 0; j < 2; ++j) {
                d[j] = fold_partials_wt(part_ptr(a.m.partials, SL_DIFF2, j));
                n[j] = fold_partials_wt(part_ptr(a.m.partials, SL_NORM2, j));
            }
        if (tid == 0) {
            int all = 1;
            for (int j = 0; j < 2; ++j) {
                const int c = (a.m.check_convergence && a.decide_check) ? (d[j] <= a.e_rel[j] * a.e_rel[j] * n[j]) : 0;
                st->conv[j] = c;
                st->norms[j][0] = d[j];
                st->norms[j][1] = n[j];
                all &= c;
                const int tau = a.m.has_prox[j] ? (j ? tau1 : tau0) : 0;
                st->sub_total[j] += tau;
                st->last_tau[j] = tau;
            }
            st->it_done += 1;
            bar->epoch[0] = ep;
            if (a.prof != nullptr) a.prof[nstamp++] = wall_clock64();
            if (a.m.check_convergence && a.decide_check && all) {
                st->stopped = 1;
                st->reason = HALT_CONVERGED;
                __threadfence();
                st->halt = 1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// block-SDMM: update of one block                       (proxmin/utils.py:295-346, nmf.py:181-185)
// ------------------------------------------------------------------------------------------------
struct BsdmmArgs {
    float* X;
    SlabRef slab;
    float* Z[PMX_MAX_G];
    float* U[PMX_MAX_G];
    int64_t rows;
    int K;
    int j;               // block
    int n_g;             // number of constraints (0: proxs_g[j] is None)
    ProxSeq prox_f;
    ProxSeq prox_g[PMX_MAX_G];
    DevStatus* status;
    double* partials;
    float* absmax_out;   // as FinishArgs: [2][EW_BLOCKS] per-workgroup max |X_j| of the block written here, or nullptr
    // User-defined operators (host round trips, one iteration per call; pmx_bsdmm_split):
    //   stage 0  the whole update on the device (no user operator)
    //   stage 1  (host_f) Tf = X - dX - step_f grad, the argument of the user's prox_f; nothing else is written
    //   stage 2  X <- prox_f(..) (device operator, or Tf as the host left it), its norms; then either the constraints
    //            (no user proxs_g member) or T_i = X + U_i for the members that are user callables
    //   stage 3  (host_g != 0) the constraints: Z_i <- T_i as the host left it (user members) or prox_g_i(X + U_i), U_i, norms
    int stage;
    int host_f;
    unsigned host_g;     // bit i: proxs_g[j][i] is a user callable
    float* Tf;
    float* T[PMX_MAX_G];
    float* gramPart;     // [r4] [2][GRAM_BLOCKS][KP*KP] or nullptr: leave the partial Gram matrices of the new X_j here (stage 0, K <= 64,
    int KP;              //      gram_per(rows) <= BSDMM_GRAM_ROWS)
};
constexpr int BSDMM_GRAM_ROWS = 64;      // rows of X_j a workgroup keeps in LDS for its partial Gram matrix (factors of <= 16384 rows)
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_bsdmm_update(BsdmmArgs a) {
    __shared__ double scratch[EW_WAVES * (2 + 4 * PMX_MAX_G)];
    __shared__ __attribute__((aligned(16))) float gtile[NC <= 2 ? BSDMM_GRAM_ROWS : 1][NC <= 2 ? 32 * NC + 4 : 4];   // the workgroup's rows of the new X_j (gramPart)
    if (chain_halted(a.status)) return;
    const int j = a.j;
    const int K = a.K;
    const float sf = (float)a.status->step[j];
    // get_step_g (utils.py:269-279): step_f * ||L||^2 * N_blocks * M_constraints, identity L
    const float sg = sf * 1.f * 2.f * (float)a.n_g;
    const float w = a.n_g > 0 ? sf / sg : 0.f;         // step_f / step_g[i]
    const float nisg = a.n_g > 0 ? -1.f / sg : 0.f;    // -1 / step_g
    const bool do_x = a.stage <= 2;                    // stages 0-2 form (or take over) the new X
    const bool do_g = a.stage == 0 || a.stage == 3 || (a.stage == 2 && a.host_g == 0u);
    float d2 = 0.f, x2 = 0.f, xmax = 0.f;
    float r2[PMX_MAX_G], s2[PMX_MAX_G], z2[PMX_MAX_G], u2[PMX_MAX_G];
#pragma unroll
    for (int i = 0; i < PMX_MAX_G; ++i) r2[i] = s2[i] = z2[i] = u2[i] = 0.f;
    // [r4] workgroup b holds the CONTIGUOUS rows per * b .. per * b + per - 1 (gram_per: k_gram_partial's shares), half-wave h of
    // it the rows h, h + 32, ..: the rows of the new X_j pass through this kernel anyway, so it leaves their partial Gram matrix
    // for the step rule of the OTHER block behind (gramPart; k_gram_partial's own summation order: bit-identical partials)
    const int l32 = threadIdx.x & 31;
    const int64_t per = gram_per(a.rows), row0 = (int64_t)blockIdx.x * per;
    const bool gram_out = NC <= 2 && a.gramPart != nullptr;
    for (int bi = 0; bi < (int)(per / 32); ++bi) {
        const int lrow = (threadIdx.x >> 5) + 32 * bi;
        const int64_t r = row0 + lrow;
        if (r >= a.rows) break;
        bool ok[NC];
        float g[NC], xo[NC], v[NC], sk[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
        if (do_x) {
            if (!(a.stage == 2 && a.host_f)) load_grad<NC>(g, ok, a.slab, a.rows, K, r, l32);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int64_t e = r * K + l32 + 32 * c;
                xo[c] = ok[c] ? a.X[e] : 0.f;
                if (a.stage == 2 && a.host_f) {
                    v[c] = ok[c] ? a.Tf[e] : 0.f;           // prox_f(..) as the host left it
                } else {
                    float dx = 0.f;
                    for (int i = 0; i < a.n_g; ++i)             // utils.py:330-336
                        if (ok[c]) dx += w * (xo[c] - a.Z[i][e] + a.U[i][e]);
                    v[c] = (xo[c] - dx) - sf * g[c];            // utils.py:338 + nmf.py:185
                }
                sk[c] = sf;
            }
            if (a.stage == 1) {                                 // the argument of the user's prox_f
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (ok[c]) a.Tf[r * K + l32 + 32 * c] = v[c];
                continue;
            }
            if (!(a.stage == 2 && a.host_f)) prox_row<NC>(v, ok, a.prox_f, sk);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (ok[c]) {
                    a.X[r * K + l32 + 32 * c] = v[c];
                    const float d = v[c] - xo[c];
                    d2 += d * d;
                    x2 += v[c] * v[c];
                    xmax = fmaxf(xmax, fabsf(v[c]));
                    if constexpr (NC <= 2) { if (gram_out) gtile[lrow][l32 + 32 * c] = v[c]; }
                }
            if (a.stage == 2 && a.host_g != 0u) {               // arguments of the user-defined members of proxs_g
                for (int i = 0; i < a.n_g; ++i)
                    if ((a.host_g >> i) & 1u) {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            if (ok[c]) { const int64_t e = r * K + l32 + 32 * c; a.T[i][e] = v[c] + a.U[i][e]; }
                    }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = ok[c] ? a.X[r * K + l32 + 32 * c] : 0.f;
        }
        if (!do_g) continue;
        for (int i = 0; i < a.n_g; ++i) {               // do_the_mm, utils.py:295-304
            float zn[NC], zo[NC], uo[NC], sgk[NC];
            const bool hosted = (a.host_g >> i) & 1u;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int64_t e = r * K + l32 + 32 * c;
                zo[c] = ok[c] ? a.Z[i][e] : 0.f;
                uo[c] = ok[c] ? a.U[i][e] : 0.f;
                zn[c] = hosted ? (ok[c] ? a.T[i][e] : 0.f) : v[c] + uo[c];
                sgk[c] = sg;
            }
            if (!hosted) prox_row<NC>(zn, ok, a.prox_g[i], sgk);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (ok[c]) {
                    const int64_t e = r * K + l32 + 32 * c;
                    const float rr = v[c] - zn[c];
                    const float sd = nisg * (zn[c] - zo[c]);
                    const float un = uo[c] + rr;
                    a.Z[i][e] = zn[c];
                    a.U[i][e] = un;
                    r2[i] += rr * rr;
                    s2[i] += sd * sd;
                    z2[i] += zn[c] * zn[c];
                    const float us = un / sg;
                    u2[i] += us * us;
                }
        }
    }
    if (a.stage == 1) return;
    if constexpr (NC <= 2) {
        if (gram_out) {                      // (uniform) partial Gram matrix of this workgroup's rows
            const int KP = a.KP;
            const int nrow = a.rows - row0 < per ? (int)(a.rows - row0 > 0 ? a.rows - row0 : 0) : (int)per;
            __syncthreads();
            float* out = a.gramPart + ((int64_t)j * GRAM_BLOCKS + blockIdx.x) * KP * KP;
            // 4 x 4 entries per thread, two 16-byte LDS reads per row and 16 products (one entry per thread read 2 words per product:
            // LDS-bound, slower than the k_gram_partial launch it replaces); every entry sums its rows in ascending order
            const int tpr = KP / 4;                              // threads per tile row
            if (nrow > 0 && (int)threadIdx.x < tpr * tpr) {
                const int ti = threadIdx.x / tpr, tj = threadIdx.x - ti * tpr;
                float acc[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) acc[i][jx] = 0.f;
                for (int rr = 0; rr < nrow; ++rr) {
                    const float4 av = *reinterpret_cast<const float4*>(&gtile[rr][4 * ti]);
                    const float4 bv = *reinterpret_cast<const float4*>(&gtile[rr][4 * tj]);
                    const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) acc[i][jx] += a4[i] * b4[jx];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gi = 4 * ti + i;
                    float4 o;
                    o.x = (gi < K && 4 * tj + 0 < K) ? acc[i][0] : 0.f;
                    o.y = (gi < K && 4 * tj + 1 < K) ? acc[i][1] : 0.f;
                    o.z = (gi < K && 4 * tj + 2 < K) ? acc[i][2] : 0.f;
                    o.w = (gi < K && 4 * tj + 3 < K) ? acc[i][3] : 0.f;
                    *reinterpret_cast<float4*>(&out[(int64_t)gi * KP + 4 * tj]) = o;
                }
            }
            __syncthreads();
        }
    }
    double red[2 + 4 * PMX_MAX_G];
    red[0] = d2;
    red[1] = x2;
#pragma unroll
    for (int i = 0; i < PMX_MAX_G; ++i) {
        red[2 + 4 * i + 0] = r2[i];
        red[2 + 4 * i + 1] = s2[i];
        red[2 + 4 * i + 2] = z2[i];
        red[2 + 4 * i + 3] = u2[i];
    }
    // slots: SL_DIFF2, SL_NORM2, then SL_G0.. ; consecutive slots are 2*EW_BLOCKS doubles apart, but
    // SL_G0 is not adjacent to SL_NORM2, so store in two groups
    if (do_x && do_g && a.n_g <= 3) {
        // [r4] the fused path with up to three constraints (2 + 4 n_g <= 14 sums): ONE transposed reduction of 16 values (wave_sum16)
        // instead of 34 butterflies -- the same sums bit for bit, a third of the instructions
        double all[16];
#pragma unroll
        for (int i = 0; i < 14; ++i) all[i] = red[i];
        all[14] = 0.0; all[15] = 0.0;
        const double tot = wave_sum16(all);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        __syncthreads();
        if (lane < 16) scratch[lane * EW_WAVES + w] = tot;
        __syncthreads();
        if ((int)threadIdx.x < 2 + 4 * a.n_g) {
            double sum = 0.0;
            for (int q = 0; q < EW_WAVES; ++q) sum += scratch[threadIdx.x * EW_WAVES + q];
            double* dst = threadIdx.x < 2 ? part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x + (int64_t)threadIdx.x * 2 * EW_BLOCKS
                                          : part_ptr(a.partials, SL_G0, j) + blockIdx.x + (int64_t)(threadIdx.x - 2) * 2 * EW_BLOCKS;
            *dst = sum;
        }
    } else {
    if (do_x) {
        double head[2] = {red[0], red[1]};
        block_sum_store<2>(head, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    }
    if (do_g) {
        double tail[4 * PMX_MAX_G];
#pragma unroll
        for (int i = 0; i < 4 * PMX_MAX_G; ++i) tail[i] = red[2 + i];
        block_sum_store<4 * PMX_MAX_G>(tail, part_ptr(a.partials, SL_G0, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    }
    }
    if (a.absmax_out != nullptr && do_x) {
        const double m = wave_max((double)xmax);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double mm = scratch[0];
            for (int q = 1; q < EW_WAVES; ++q) mm = fmax(mm, scratch[q]);
            a.absmax_out[j * EW_BLOCKS + blockIdx.x] = (float)mm;
        }
    }
}

// check_constraint_convergence for block j (utils.py:349-391), and end-of-iteration bookkeeping
struct BsdmmDecideArgs {
    DevStatus* status;
    double* partials;
    int j;
    int n_g;
    int64_t size;        // X.size == Z.size (identity L)
    double e_rel, e_abs;
    int last_block;      // 1 for the second block: closes the iteration (algorithms.py:841-844)
    const float* comm_scalars;   // row-sharded block A: all-reduced sums [d2, x2, q0..] instead of the local partials
};
// (body: also the extra workgroup of k_gram_reduce, 256 threads -- fold_partials is per wave, any workgroup size)
__device__ __forceinline__ void bsdmm_decide_body(const BsdmmDecideArgs& a, double* scratch) {
    const int j = a.j;
    double d2, x2, q[4 * PMX_MAX_G];
    if (a.comm_scalars != nullptr) {
        d2 = (double)a.comm_scalars[0];
        x2 = (double)a.comm_scalars[1];
        for (int i = 0; i < 4 * a.n_g; ++i) q[i] = (double)a.comm_scalars[2 + i];
    } else {
        d2 = fold_partials(part_ptr(a.partials, SL_DIFF2, j), scratch);
        x2 = fold_partials(part_ptr(a.partials, SL_NORM2, j), scratch);
        for (int i = 0; i < 4 * a.n_g; ++i) q[i] = fold_partials(part_ptr(a.partials, SL_G0 + i, j), scratch);
    }
    if (threadIdx.x == 0) {
        DevStatus* st = a.status;
        const double sq = sqrt((double)a.size);
        int conv = 1;
        if (a.n_g == 0) {
            // utils.py:319-327: R = 0, S = X_new - X_old, Z = X, U = 0, step_g = None
            const double e_pri = sq * a.e_abs + a.e_rel * sqrt(x2);
            const double e_dual = sq * a.e_abs + a.e_rel * 0.0;
            conv = (0.0 <= e_pri) && (sqrt(d2) <= e_dual);
        } else {
            for (int i = 0; i < a.n_g; ++i) {
                const double lR = sqrt(q[4 * i + 0]), lS = sqrt(q[4 * i + 1]);
                const double e_pri = sq * a.e_abs + a.e_rel * fmax(sqrt(x2), sqrt(q[4 * i + 2]));
                const double e_dual = sq * a.e_abs + a.e_rel * sqrt(q[4 * i + 3]);
                conv &= (lR <= e_pri) && (lS <= e_dual);
            }
        }
        st->conv[j] = conv;
        st->norms[j][0] = d2;
        st->norms[j][1] = x2;
        if (a.last_block) {
            st->it_done += 1;
            if (st->conv[0] && st->conv[1]) {
                st->stopped = 1;
                st->reason = HALT_CONVERGED;
                __threadfence();
                st->halt = 1;
            }
        }
    }
}
__global__ __launch_bounds__(EW_THREADS) void k_bsdmm_decide(BsdmmDecideArgs a) {
    __shared__ double scratch[EW_WAVES];
    if (chain_halted(a.status)) return;
    bsdmm_decide_body(a, scratch);
}

// ------------------------------------------------------------------------------------------------
// row-sharded multi-GPU: pack what needs a cross-rank sum into the comm buffer, and consume it
//   comm = [ gSt (N*K) | Gram(A) (KP*KP) | colsum(A) (MAXK) | scalars (32) ]      (float32)
//   scalars: [0] = sum (A_new - A_old)^2, [1] = sum A_new^2 of the PREVIOUS iteration (outer test)
// ------------------------------------------------------------------------------------------------
struct PackArgs {
    SlabRef slabS;           // local gSt slabs
    float* comm;
    int64_t N;
    int K, KP;
    const double* colpart;   // block 0 partial column sums of the local A rows
    const double* partials;  // SL_DIFF2 / SL_NORM2 of block 0 (local rows)
    const double* gramA;     // local A^T A (KP*KP doubles) or nullptr
    const DevStatus* status;
    int fold_grad;           // 0: only the extras (final convergence flush)
    int n_extra;             // bsdmm: additional block-0 slots SL_G0 .. SL_G0+n_extra-1 -> scalars[2..]
};
// scalars[N_SCALARS - 1] of the comm buffer is the ranks' collective halt flag: a rank whose chain is halted (converged,
// out of proximal passes, a fault) still runs this one store, the all-reduce sums it, and k_shard_post / k_shard_gram_in
// stop EVERY rank before the update of that iteration -- ranks can never disagree about which iterations were applied,
// so they always issue the same number of collectives (a rank-local decision there would be a hang, not an error).
constexpr int SHARD_HALT_SLOT = 31;
// [r4] Grid = PACK_XWG "extras" workgroups in FRONT of the EW_BLOCKS that fold the gS slabs: the small sums are chains of dependent
// round trips (column-sum fold, stopping sums, Gram copy) that workgroup 0 used to run BEHIND its share of the rows -- the whole
// launch waited on that one workgroup (25 us for 25 MB of traffic); now they run beside the fold, one piece per workgroup.
constexpr int PACK_XWG = 3;
// four (or fewer) partial-sum folds with every load in flight before the first add; each sum in fold_partials' own order
template <int NS>
__device__ __forceinline__ void fold_partials_many(const double* const (&part)[NS], double (&out)[NS]) {
    const int lane = threadIdx.x & 63;
    double v[NS][EW_BLOCKS / 64];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < EW_BLOCKS / 64; ++i) v[q][i] = part[q][i * 64 + lane];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < EW_BLOCKS / 64; ++i) t += v[q][i];
        out[q] = wave_sum(t);
    }
}
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_shard_pack(PackArgs a) {
    __shared__ double scratch[EW_WAVES];
    if (chain_halted(a.status)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.comm[a.N * a.K + a.KP * a.KP + MAXK + SHARD_HALT_SLOT] = 1.f;
        return;
    }
    const int K = a.K;
    if (blockIdx.x >= PACK_XWG) {
        if (a.fold_grad) {
            const int l32 = threadIdx.x & 31;
            const int64_t hw_ = ((int64_t)(blockIdx.x - PACK_XWG) * EW_THREADS + threadIdx.x) >> 5;
            const int64_t nhw_ = ((int64_t)EW_BLOCKS * EW_THREADS) >> 5;
            for (int64_t r = hw_; r < a.N; r += nhw_) {
                bool ok[NC];
                float g[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
                load_grad<NC>(g, ok, a.slabS, a.N, K, r, l32);
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (ok[c]) a.comm[r * K + l32 + 32 * c] = g[c];
            }
        }
        return;
    }
    float* ex = a.comm + a.N * K;
    float* cs = ex + a.KP * a.KP;
    float* sc = cs + MAXK;
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {               // Gram(A) (pgm / bsdmm) or zeros
        if (a.gramA != nullptr)
            for (int e = t; e < a.KP * a.KP; e += EW_THREADS) ex[e] = (float)a.gramA[e];
        else
            for (int e = t; e < a.KP * a.KP; e += EW_THREADS) ex[e] = 0.f;
    } else if (blockIdx.x == 1) {        // column sums of the local rows of A
        __shared__ double asum[ALPHA_NG][MAXK];
        colsum_fold(a.colpart, K, false, asum);   // block 0
        if (t < MAXK) {
            double s = 0.0;
            for (int q = 0; q < ALPHA_NG; ++q) s += asum[q][t];
            cs[t] = (float)s;
        }
    } else {                             // stopping sums of block 0 (+ bsdmm's extra slots); [31] = the halt flag: 0 here
        const double* const two[2] = {a.partials + ((int64_t)SL_DIFF2 * 2 + 0) * EW_BLOCKS, a.partials + ((int64_t)SL_NORM2 * 2 + 0) * EW_BLOCKS};
        double dn[2];
        fold_partials_many<2>(two, dn);
        if (t < 32) sc[t] = t == 0 ? (float)dn[0] : (t == 1 ? (float)dn[1] : 0.f);
        for (int i = 0; i < a.n_extra; ++i) {
            const double q = fold_partials(a.partials + ((int64_t)(SL_G0 + i) * 2 + 0) * EW_BLOCKS, scratch);
            if (t == 0) sc[2 + i] = (float)q;
        }
    }
}

// all-reduced Gram(A) (float32 in the comm buffer) -> the fp64 matrix k_eig reads (factor 0)
struct GramInArgs {
    const float* comm_gram;
    double* G;               // gramG[0]
    int n;
    const DevStatus* status;
    const float* peer_halt;  // the all-reduced halt flag (bsdmm: this is the first kernel after the collective) or nullptr
    DevStatus* wstatus;
};
__global__ __launch_bounds__(256) void k_shard_gram_in(GramInArgs a) {
    if (chain_halted(a.status)) return;
    if (a.peer_halt != nullptr && *a.peer_halt > 0.5f) {     // another rank is halted: stop here, before this iteration's update
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.wstatus->reason = HALT_PEER;
            __threadfence();
            a.wstatus->halt = 1;
        }
        return;
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < a.n) a.G[e] = (double)a.comm_gram[e];
}

// after the all-reduce: global step sizes for A and the deferred outer test of the previous iteration
struct ShardPostArgs {
    AlphaArgs al;            // comm_colsum set
    const float* scalars;    // comm scalars
    double* partials;        // local S-block sums
    double e_rel[2];
    int check_convergence;
    int have_prev;           // 0 on the first iteration (nothing to test yet)
    int do_alpha;            // adaprox only
};
__global__ __launch_bounds__(EW_THREADS) void k_shard_post(ShardPostArgs a) {
    __shared__ double scratch[EW_WAVES];
    DevStatus* st = a.al.status;
    if (chain_halted(st)) return;
    if (a.scalars[SHARD_HALT_SLOT] > 0.5f) {         // another rank is halted: stop before this iteration's update
        if (threadIdx.x == 0) {
            st->reason = HALT_PEER;
            __threadfence();
            st->halt = 1;
        }
        return;
    }
    if (a.do_alpha) compute_alpha(a.al);
    if (a.check_convergence && a.have_prev) {
        const double dS = fold_partials(part_ptr(a.partials, SL_DIFF2, 1), scratch);
        const double nS = fold_partials(part_ptr(a.partials, SL_NORM2, 1), scratch);
        if (threadIdx.x == 0) {
            const double dA = (double)a.scalars[0], nA = (double)a.scalars[1];
            const int cA = dA <= a.e_rel[0] * a.e_rel[0] * nA;
            const int cS = dS <= a.e_rel[1] * a.e_rel[1] * nS;
            st->conv[0] = cA;
            st->conv[1] = cS;
            st->norms[0][0] = dA; st->norms[0][1] = nA;
            st->norms[1][0] = dS; st->norms[1][1] = nS;
            if (cA && cS) {
                st->stopped = 1;
                st->reason = HALT_CONVERGED;
                __threadfence();
                st->halt = 1;
            }
        }
    }
}

// ---- S-split (pmx_set_s_split): the comm buffer is `world` chunks, chunk q = [ gSt rows of rank q | Gram | colsum(A) |
// colsum(S) | scalars ]; the small sums are copied into every chunk (a reduce-scatter hands each rank ONE chunk, summed) ----
struct PackSplitArgs {
    SlabRef slabS;           // local gSt slabs (all N rows)
    float* comm;
    int64_t N;
    int K, KP;
    const double* colpart;   // [2][EW_BLOCKS][MAXK]: block 0 = local rows of A, block 1 = this rank's columns of S
    const double* partials;  // SL_DIFF2 / SL_NORM2 of both blocks (local rows / own columns)
    const DevStatus* status;
    int fold_grad;
    int world;
    int64_t sncol, chunk;
    int zero_gram;           // write zeros into every chunk's (unused) Gram section: the first pack into a buffer only
    const double* gramA;     // [r4] pgm: the local A^T A (KP*KP doubles) into every chunk's Gram section, every iteration; nullptr: adaprox
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k_shard_pack_split(PackSplitArgs a) {
    const int64_t ex0 = a.sncol * a.K;                   // offset of the extras inside a chunk
    const int64_t sc0 = ex0 + (int64_t)a.KP * a.KP + 2 * MAXK;
    if (chain_halted(a.status)) {
        if (blockIdx.x == 0 && threadIdx.x < a.world) a.comm[threadIdx.x * a.chunk + sc0 + SHARD_HALT_SLOT] = 1.f;
        return;
    }
    const int K = a.K;
    if (blockIdx.x >= PACK_XWG) {            // the fold of the gS slabs into the ranks' chunks (extras: the workgroups in front)
        if (a.fold_grad) {
            const int l32 = threadIdx.x & 31;
            const int64_t hw_ = ((int64_t)(blockIdx.x - PACK_XWG) * EW_THREADS + threadIdx.x) >> 5;
            const int64_t nhw_ = ((int64_t)EW_BLOCKS * EW_THREADS) >> 5;
            for (int64_t r = hw_; r < a.N; r += nhw_) {
                bool ok[NC];
                float g[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
                load_grad<NC>(g, ok, a.slabS, a.N, K, r, l32);
                const int q = (int)r / (int)a.sncol;             // (32-bit: a 64-bit division per row made this kernel twice as long)
                float* dst = a.comm + (int64_t)q * a.chunk + (int64_t)((int)r - q * (int)a.sncol) * K;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (ok[c]) dst[l32 + 32 * c] = g[c];
            }
        }
        return;
    }
    const int t = threadIdx.x;
    if (blockIdx.x < 2) {                    // column sums of block j = blockIdx.x (local rows of A / own columns of S), into every chunk
        const int j = blockIdx.x;
        __shared__ double asum[ALPHA_NG][MAXK];
        colsum_fold(a.colpart + (int64_t)j * EW_BLOCKS * MAXK, K, false, asum);
        if (t < MAXK) {
            double s = 0.0;
            for (int q = 0; q < ALPHA_NG; ++q) s += asum[q][t];
            const float v = (float)s;
            for (int q = 0; q < a.world; ++q) a.comm[q * a.chunk + ex0 + (int64_t)a.KP * a.KP + j * MAXK + t] = v;
        }
        return;
    }
    // stopping sums of both blocks (+ the halt flag's zero), into every chunk.  The Gram section of a chunk (KP^2 floats: the
    // layout's, unused by adaprox) is zeroed when the host says so (first pack after pmx_set_comm_buffer): nothing writes it
    // afterwards, and one workgroup storing world x 64 KB of zeros per iteration was a third of this launch
    const double* const four[4] = {a.partials + ((int64_t)SL_DIFF2 * 2 + 0) * EW_BLOCKS, a.partials + ((int64_t)SL_NORM2 * 2 + 0) * EW_BLOCKS,
                                   a.partials + ((int64_t)SL_DIFF2 * 2 + 1) * EW_BLOCKS, a.partials + ((int64_t)SL_NORM2 * 2 + 1) * EW_BLOCKS};
    double sums[4];
    fold_partials_many<4>(four, sums);
    for (int q = 0; q < a.world; ++q) {
        float* ex = a.comm + q * a.chunk + ex0;
        if (a.gramA != nullptr)
            for (int e = t; e < a.KP * a.KP; e += EW_THREADS) ex[e] = (float)a.gramA[e];
        else if (a.zero_gram)
            for (int e = t; e < a.KP * a.KP; e += EW_THREADS) ex[e] = 0.f;
        if (t < 32) ex[a.KP * a.KP + 2 * MAXK + t] = t < 4 ? (float)sums[t] : 0.f;
    }
}
// after the reduce-scatter: global step sizes of BOTH blocks and the deferred outer test of the previous iteration
struct ShardPostSplitArgs {
    DevStatus* status;
    const float* extras;     // this rank's chunk past gSt and the Gram matrix: colsum(A) | colsum(S) | scalars
    int64_t rows_global[2];
    int K;
    int use_fixed;
    float fixed[2];
    double e_rel[2];
    int check_convergence;
    int have_prev;
};
__global__ __launch_bounds__(EW_THREADS) void k_shard_post_split(ShardPostSplitArgs a) {
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    const float* sc = a.extras + 2 * MAXK;
    if (sc[SHARD_HALT_SLOT] > 0.5f) {              // another rank is halted: stop before this iteration's update
        if (threadIdx.x == 0) {
            st->reason = HALT_PEER;
            __threadfence();
            st->halt = 1;
        }
        return;
    }
    const int t = threadIdx.x;
    if (a.use_fixed != 2 && t < a.K)
        for (int j = 0; j < 2; ++j)
            st->alpha[j][t] = a.use_fixed ? a.fixed[j] : (float)((double)a.extras[j * MAXK + t] / (double)a.rows_global[j]) / 10.f;
    if (a.check_convergence && a.have_prev && t == 0) {
        const double dA = (double)sc[0], nA = (double)sc[1], dS = (double)sc[2], nS = (double)sc[3];
        const int cA = dA <= a.e_rel[0] * a.e_rel[0] * nA;
        const int cS = dS <= a.e_rel[1] * a.e_rel[1] * nS;
        st->conv[0] = cA;
        st->conv[1] = cS;
        st->norms[0][0] = dA; st->norms[0][1] = nA;
        st->norms[1][0] = dS; st->norms[1][1] = nS;
        if (cA && cS) {
            st->stopped = 1;
            st->reason = HALT_CONVERGED;
            __threadfence();
            st->halt = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launch wrappers (NC dispatch)
// ------------------------------------------------------------------------------------------------
#define DISPATCH_NC(K, KERNEL, grid, stream, args)                                               \
    do {                                                                                         \
        if ((K) <= 32) hipLaunchKernelGGL(KERNEL<1>, grid, dim3(EW_THREADS), 0, stream, args);    \
        else if ((K) <= 64) hipLaunchKernelGGL(KERNEL<2>, grid, dim3(EW_THREADS), 0, stream, args); \
        else hipLaunchKernelGGL(KERNEL<4>, grid, dim3(EW_THREADS), 0, stream, args);              \
    } while (0)

void launch_fold(const FoldArgs& a, int nblocks_y, hipStream_t s) { DISPATCH_NC(a.K, k_fold, dim3(EW_BLOCKS, nblocks_y), s, a); }
void launch_prox_apply(const ProxArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_prox_apply, dim3(EW_BLOCKS), s, a); }
void launch_pgm_update(const PgmArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_pgm_update, dim3(a.nbx > 0 ? a.nbx : EW_BLOCKS, 2), s, a); }
void launch_bb_reduce(const BBArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_bb_reduce, dim3(EW_BLOCKS, 2), s, a); }
void launch_bb_step(const BBStepArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_bb_step, dim3(1), dim3(EW_THREADS), 0, s, a); }
void launch_bt_update(const BtArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_bt_update, dim3(EW_BLOCKS, 2), s, a); }
void launch_bt_collect(const BtCollectArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_bt_collect, dim3(1), dim3(EW_THREADS), 0, s, a); }
void launch_bt_finish(const BtFinishArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_bt_finish, dim3(EW_BLOCKS, 2), dim3(EW_THREADS), 0, s, a); }
void launch_pgm_decide(const DecideArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_pgm_decide, dim3(1), dim3(EW_THREADS), 0, s, a); }
void launch_colsum(const ColsumArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_colsum, dim3(EW_BLOCKS, 2), s, a); }
void launch_alpha_init(const AlphaArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_alpha_init, dim3(1), dim3(EW_THREADS), 0, s, a); }
void launch_colscale(const ColScaleArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_colscale, dim3(EW_BLOCKS), s, a); }
void launch_ada_moment(const MomentArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_ada_moment, dim3(EW_BLOCKS, 2), s, a); }
#define DISPATCH_NC_NT(K, NT, KERNEL, grid, stream, args)                                                          \
    do {                                                                                                           \
        if ((K) <= 32) hipLaunchKernelGGL((KERNEL<1, NT>), grid, dim3(EW_THREADS), 0, stream, args);                \
        else if ((K) <= 64) hipLaunchKernelGGL((KERNEL<2, NT>), grid, dim3(EW_THREADS), 0, stream, args);           \
        else hipLaunchKernelGGL((KERNEL<4, NT>), grid, dim3(EW_THREADS), 0, stream, args);                          \
    } while (0)
void launch_ada_sub(const SubArgs& a, hipStream_t s) {
    if (a.nt == 1) DISPATCH_NC_NT(a.K, 1, k_ada_sub, dim3(EW_BLOCKS, 2), s, a);
    else if (a.nt == 4) DISPATCH_NC_NT(a.K, 4, k_ada_sub, dim3(EW_BLOCKS, 2), s, a);
    else DISPATCH_NC_NT(a.K, SUB_NT_MAX, k_ada_sub, dim3(EW_BLOCKS, 2), s, a);
}
void launch_ada_finish(const FinishArgs& a, hipStream_t s) { DISPATCH_NC(a.s.K, k_ada_finish, dim3(EW_BLOCKS, 2), s, a); }
void launch_ada_decide(const AdaDecideArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_ada_decide, dim3(1), dim3(EW_THREADS), 0, s, a); }
// dynamic LDS of the fused tail: three state arrays of (slots[0] + slots[1]) x NC x 1024 floats
static size_t ada_tail_lds_bytes(const TailArgs& a) {
    const int NC = a.m.K <= 32 ? 1 : (a.m.K <= 64 ? 2 : 4);
    const size_t arr = (size_t)(a.slots[0] + a.slots[1]) * NC * EW_THREADS * sizeof(float);
    return 2 * arr + std::max(arr, (size_t)(EW_THREADS / 32) * MAXK * sizeof(float));
}
constexpr size_t ADA_TAIL_LDS_MAX = 160 * 1024 - 6 * 1024;   // the kernel's static scratch is ~3.2 KB
template <int NC>
static hipError_t launch_ada_tail_t(const TailArgs& a, hipStream_t s) {
    const size_t lds = ada_tail_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void*)k_ada_tail<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ada_tail<NC>, dim3(EW_BLOCKS), dim3(EW_THREADS), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_ada_tail(const TailArgs& a, hipStream_t s) {
    if (a.m.K <= 32) return launch_ada_tail_t<1>(a, s);
    if (a.m.K <= 64) return launch_ada_tail_t<2>(a, s);
    return launch_ada_tail_t<4>(a, s);
}
void launch_bsdmm_update(const BsdmmArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_bsdmm_update, dim3(EW_BLOCKS), s, a); }
void launch_shard_pack(const PackArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_shard_pack, dim3(EW_BLOCKS + PACK_XWG), s, a); }
void launch_shard_gram_in(const GramInArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_shard_gram_in, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }
void launch_shard_post(const ShardPostArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_shard_post, dim3(1), dim3(EW_THREADS), 0, s, a); }
void launch_shard_pack_split(const PackSplitArgs& a, hipStream_t s) { DISPATCH_NC(a.K, k_shard_pack_split, dim3(EW_BLOCKS + PACK_XWG), s, a); }
void launch_shard_post_split(const ShardPostSplitArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_shard_post_split, dim3(1), dim3(EW_THREADS), 0, s, a); }
void launch_bsdmm_decide(const BsdmmDecideArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_bsdmm_decide, dim3(1), dim3(EW_THREADS), 0, s, a); }
