// Internal definitions shared by the gfx950 kernels and the C-ABI layer of libpmx.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pmx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------
// elementwise / reduction grid: every factor-sized kernel uses the same fixed geometry so that
// per-block partial sums written by one kernel can be folded by the next one in a fixed order
// (deterministic, no float atomics).
// ------------------------------------------------------------------------------------------
constexpr int EW_BLOCKS = 256;   // one per CU, 16 wavefronts each: factor-sized kernels are latency-bound
constexpr int EW_THREADS = 1024;
constexpr int EW_WAVES = EW_THREADS / 64;
constexpr int GRAM_BLOCKS = 256;          // [r4] 256 (was 128): one share per workgroup of the update kernels (EW_BLOCKS), see gram_per
// Rows of a factor per partial Gram matrix: contiguous shares of a multiple of 32 rows, the same for k_gram_partial and for the
// update kernels that leave the partials of their own rows behind (k_pgm_update, k_bsdmm_update: workgroup b holds rows
// per * b .. per * b + per - 1) -- so the step rule is the same number bit for bit whichever kernel summed the rows.
__host__ __device__ inline int64_t gram_per(int64_t rows) { return 32 * ((rows + (int64_t)GRAM_BLOCKS * 32 - 1) / ((int64_t)GRAM_BLOCKS * 32)); }
__host__ __device__ inline int gram_nparts(int64_t rows) { const int64_t per = gram_per(rows); return (int)((rows + per - 1) / per); }

constexpr int LPR = 32;          // lanes per row (half a wavefront): K <= 128 -> <= 4 values per lane
constexpr int MAXC = 4;          // ceil(128 / LPR)
constexpr int MAXK = 128;

constexpr int SUB_RING = 16;     // passes whose stopping-test sums are kept (two launches of SUB_NT_MAX passes)
constexpr int SUB_NT_MAX = 8;    // passes of the proximal sub-iteration one launch runs back to back
// reduction slots (double per block), indexed [slot][block(A|S)][EW_BLOCKS]
enum {
    SL_DIFF2 = 0,   // sum (X_new - X_old)^2            (algorithms.py:131,406)
    SL_NORM2 = 1,   // sum X_new^2
    SL_MAXPSI = 6,  // max Psi                          (algorithms.py:384)
    SL_G0 = 8,      // bsdmm: 5 sums per constraint i: R^2, Sd^2, Z^2, (U/sg)^2 ; + X^2 in SL_NORM2
    SL_BB0 = 8 + 4 * PMX_MAX_G,   // Barzilai-Borwein: sum s^2, sum s.y, sum y^2, sum g^2, max|x|, max|g|  (utils.py:216-241)
    SL_BT0 = 8 + 4 * PMX_MAX_G + 6,   // backtracking: sum (X-X_).G, max|G|, max|X_|        (algorithms.py:117-121)
    // adaprox proximal sub-iterations: per pass t the two sums of the stopping test (algorithms.py:389),
    // slot SL_SUBR0 + 2 * (t % SUB_RING) = sum (z_new - z)^2, + 1 = sum z^2
    SL_SUBR0 = 8 + 4 * PMX_MAX_G + 6 + 3,
    SL_COUNT = SL_SUBR0 + 2 * 16
};
constexpr int COLSUM_SLOTS = MAXK;   // per-block per-component partial column sums

// device-resident control block.  Every kernel of a chain starts by reading `halt`.
struct DevStatus {
    int halt;            // != 0: remaining kernels of the chain are no-ops
    int reason;          // why (HALT_*)
    int it_done;         // completed iterations since *_begin
    int conv[2];         // last outer convergence flags
    int stopped;         // outer test fired
    int sub_done[2];     // adaprox: sub-iteration loop finished for block j in this iteration
    int sub_tau[2];      // adaprox: tau reached in this iteration
    int need_sub[2];     // adaprox: k_ada_finish found the loop unfinished (more passes must be enqueued)
    int last_tau[2];     // adaprox: tau of the last completed iteration (host uses it to size the next chain)
    long long sub_total[2];
    double step[2];      // pgm/bsdmm: current step sizes
    double lam[2];       // largest Gram eigenvalues
    double maxpsi[2];
    double loss;
    double norms[2][2];  // [block][diff2,norm2] of the last outer test
    float alpha[2][MAXK];   // adaprox per-component steps (nmf.py:93)
    float gamma[2][MAXK];   // alpha / max(Psi)
    float ratio[2][MAXK];   // gamma / alpha  (NaN when alpha == 0, like the reference)
    double bb_delta[2];     // Barzilai-Borwein stabilisation radius (utils.py:237-239)
    double bt[2][5];        // backtracking sums per block: (X-X_).G, (X-X_)^2, max|G|, max|X_|, X^2
    double eigvec[2][MAXK]; // warm start for the power iteration
    int eig_iters[2];
    int tail_fault;      // k_ada_tail: its census barrier found the workgroups not co-resident; nothing was written (host falls back to the separate kernels)
    int pad2;
    int k1_fault;        // k_grad_f16_v8<CHAIN>: 1 a chain predecessor never arrived, 2 it runs on another XCD (host falls back to slabs);
                         // (3: injected by the tests) [r4] 4: a two-term fp16 K1 found the residual's bound too far above max|Y| for ONE fp16 scale (host falls back to exact fp32)
};
enum { HALT_NONE = 0, HALT_CONVERGED = 1, HALT_NEED_SUB = 2, HALT_ERROR = 3,
       HALT_RETRY = 4,   // (host view) a kernel reported a recoverable fault before anything was updated: re-enqueue from it_done
       HALT_PEER = 5 };  // row-sharded runs: another rank's chain is halted; this one stopped at the same iteration

struct ProxSeq {           // device copy of pmx_proxseq
    int n, repeat;
    pmx_prox seq[PMX_MAX_SEQ];
};

__device__ __forceinline__ bool chain_halted(const DevStatus* st) {
    return __builtin_nontemporal_load(&st->halt) != 0;
}

// ------------------------------------------------------------------------------------------
// [r6] The step rule's Gram fold (k_gram_reduce's work) riding in the first workgroups of the NEXT K1 launch.
// pgm: Gram(X_t) is needed by the update kernel of iteration t only (step = 1 / lmax), not by K1; its partials are left by the update
// kernel of iteration t - 1.  As a launch of its own the fold is 5.5 us of a 50 us iteration at 4096 x 4096 x 32 (pure latency: two
// round trips to memory); K1's workgroups open with round trips of their own that the fold's ride along with.  Same arithmetic, same
// order as k_gram_reduce (eight threads fold up to 32 partials each, two batches of 16 loads in flight; the eight sums in a fixed
// order): the step rule is the same number bit for bit whichever kernel folded.  The stopping test of iteration t - 1 rides in one
// more workgroup, as it does in k_gram_reduce.
// ------------------------------------------------------------------------------------------
struct K1GramFold {
    const float* part;       // [2][GRAM_BLOCKS][KP*KP]; nullptr: nothing rides in this launch
    double* G;               // [2][KP*KP]
    int KP;
    int nparts[2];
    double* dec_partials;    // the previous iteration's stopping test (nullptr: none)
    DevStatus* dec_status;
    double dec_e_rel[2];
};
__device__ __forceinline__ void pgm_decide_body(DevStatus* st, double* partials, const double (&e_rel)[2], int check, bool wt);   // k_update.hip
// t: index of the calling thread among the 8 * KP * KP threads that fold factor f (octet e = t >> 3 owns entry e)
__device__ __forceinline__ void gram_fold_octet(const float* part, double* G, int KP, int np, int f, int t) {
    const int n = KP * KP, e = t >> 3, q = t & 7;
    double s = 0.0;
    if (e < n) {
        const float* p = part + (int64_t)f * GRAM_BLOCKS * n + (int64_t)(q * 32) * n + e;
        const int mine = np - q * 32 < 32 ? np - q * 32 : 32;          // partials of this thread (<= 0: none)
        for (int b0 = 0; b0 < mine; b0 += 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = p[(int64_t)(b0 + i < mine ? b0 + i : mine - 1) * n];   // (unconditional loads: all 16 in flight)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += b0 + i < mine ? (double)v[i] : 0.0;
        }
    }
    const int base = threadIdx.x & 56;       // (an octet never straddles a wave; every lane takes part in the shuffles)
    double tot = __shfl(s, base);
#pragma unroll
    for (int i = 1; i < 8; ++i) tot += __shfl(s, base + i);
    if (e < n && q == 0) G[(int64_t)f * n + e] = tot;
}
// workgroups the fold occupies in a launch of `threads`-thread workgroups (+ 1 for the stopping test)
__host__ __device__ inline int k1_gram_fold_wgs(int KP, int threads) { return 2 * ((8 * KP * KP + threads - 1) / threads) + 1; }
// called by every workgroup of K1 before anything else (uniform per workgroup); the workgroup then goes on with its region
__device__ __forceinline__ void k1_gram_fold(const K1GramFold& g) {
    if (g.part == nullptr) return;
    const int per = (8 * g.KP * g.KP + (int)blockDim.x - 1) / (int)blockDim.x, b = blockIdx.x;
    if (b < 2 * per) {
        const int f = b / per;
        gram_fold_octet(g.part, g.G, g.KP, g.nparts[f], f, (b - f * per) * (int)blockDim.x + (int)threadIdx.x);
    } else if (b == 2 * per && g.dec_partials != nullptr) {
        pgm_decide_body(g.dec_status, g.dec_partials, g.dec_e_rel, 1, false);
    }
}

// tall factor descriptor: X is rows x K, row-major, K contiguous
struct Tall {
    float* p;
    int64_t rows;
};

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            pmx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PMX_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)

void pmx_set_error(const char* fmt, ...);
