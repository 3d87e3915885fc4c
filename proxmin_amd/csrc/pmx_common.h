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
