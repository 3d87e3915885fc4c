// libpmx: context management, kernel chains and the C ABI declared in include/pmx.h.
//
// Host-side mirror of the reference's solver loops (proxmin/algorithms.py pgm :87-138,
// adaprox :365-413, bsdmm :800-844) as *kernel chains*: the host enqueues whole iterations on one HIP
// stream without synchronising; convergence tests and the data-dependent length of adaprox's proximal
// sub-iteration loop are resolved on the device through DevStatus (see pmx_common.h).  The host only
// synchronises at the end of a chunk of iterations, reads DevStatus back, and resumes if a chain
// stopped early.
//
// Single translation unit: the kernels are included below.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>
#include <mutex>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <dlfcn.h>
#include <link.h>
#include "pmx_common.h"
#include "chain_link.h"
#include "k_grad.hip"
#include "k_grad_bf16.hip"
#include "k_grad_k128.hip"
#include "k_grad_f16_k32.hip"
#include "k_grad_f32pc.hip"
#include "k_update.hip"
#include "k_gram.hip"
#include "k_gfix.hip"
#include "k_small_f64.hip"
#include "k_grad_f64.hip"
#include "k_big_f64.hip"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void pmx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define FAIL(code, ...)            \
    do {                           \
        pmx_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)

enum Algo { ALG_NONE = 0, ALG_PGM, ALG_ADAPROX, ALG_BSDMM };

struct pmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t M = 0, N = 0, K = 0;
    int KP = 0, mode = PMX_MODE_F32;
    int64_t rows[2] = {0, 0};

    // data
    const float* Y = nullptr;
    float* Yown = nullptr;
    int64_t ldY = 0;
    bool haveY = false;
    // K1's FRAME (choose_frame): M x N, or -- a ragged shape whose K has a producer / consumer kernel -- M and N rounded up to that
    // kernel's tile.  Y (and W) then live in a zero-padded copy of the frame's size, the factor arrays K1 reads and the gradient
    // slabs it writes have the frame's rows (the extra ones zero / never read); everything else works on the M and N real rows.
    int64_t Mk = 0, Nk = 0;
    int64_t rowsK[2] = {0, 0};
    bool framed = false;
    // ... and K: a K that has no tuned K1 (anything but 32 / 64 / 128) runs the next one's (Kk) on COPIES of the factors with Kk
    // floats per row, zero behind column K (k_pad_factors in front of every K1 launch); the gradient slabs have Kk floats per row
    // (SlabRef::ld), the folds read the first K of them.  Everything outside K1 keeps the real K and its own arrays.
    int64_t Kk = 0;
    float* Xk[2] = {nullptr, nullptr};

    float* X[2] = {nullptr, nullptr};      // A, St
    float* G[2] = {nullptr, nullptr};
    float* Xe[2] = {nullptr, nullptr};     // pgm accelerated
    float* Xp[2] = {nullptr, nullptr};     // adaprox X_
    float* Mm[2] = {nullptr, nullptr};
    float* Vv[2] = {nullptr, nullptr};
    float* Vh[2] = {nullptr, nullptr};
    float* Psi[2] = {nullptr, nullptr};
    float* zb[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float* Zg[2][PMX_MAX_G] = {};
    float* Ug[2][PMX_MAX_G] = {};
    float* bbX[2] = {nullptr, nullptr};    // Barzilai-Borwein X_prev
    float* bbG[2] = {nullptr, nullptr};    // Barzilai-Borwein G_prev

    // PMX_MODE_F64 (small problems, pgm / FISTA: k_small_f64.hip): the context's arrays in fp64; none of the float arrays above exist
    bool f64 = false;
    double* Yd = nullptr;
    double* Xd[2] = {nullptr, nullptr};
    double* Xed[2] = {nullptr, nullptr};
    double* Gd[2] = {nullptr, nullptr};
    double* slabd[2] = {nullptr, nullptr};
    double* Md[2] = {nullptr, nullptr};    // [r4] adaprox in fp64 (k64_ada_iter): moments, running maximum, X_, Psi, z, the steps of the last iteration
    double* Vd[2] = {nullptr, nullptr};
    double* Vhd[2] = {nullptr, nullptr};
    double* Xpd[2] = {nullptr, nullptr};
    double* Psid[2] = {nullptr, nullptr};
    double* zd[2] = {nullptr, nullptr};
    double* alpha64 = nullptr;
    double* Zd[2][PMX_MAX_G] = {};         // [r4] bsdmm in fp64 (k64_bsdmm_block)
    double* Ud[2][PMX_MAX_G] = {};
    int t64x = 0, t64y = 0;                // K1 tiles: column tiles (-> gA slabs), row tiles (-> gSt slabs)
    // [r6] fp64 at any size (k_big_f64.hip): the same context, other launches
    bool f64big = false;
    double* gramPart64 = nullptr;          // [2][G64_BLOCKS][KP*KP]
    double* colpart64 = nullptr;           // [2][EW_BLOCKS][MAXK]
    int nsplit64[2] = {1, 1}, bps64[2] = {1, 1};   // sweep plan of the gradient pass of block j (0: gA, fixed factor A; 1: gSt, fixed factor St)
    int nsub64 = 4;                        // adaprox: proximal passes enqueued per iteration (follows the loops' lengths)
    bool Wd_on = false;
    double* Xprevd[2] = {nullptr, nullptr}; // pgm line search in fp64: X_ (algorithms.py:102)
    double* Wd = nullptr;                  // weights of the likelihood (pmx_set_W_host_f64), Yd's shape and pitch; nullptr: W == 1
    double* Xk64[2] = {nullptr, nullptr};  // K1's padded operands (ceil64(rows) x KP), when the factors are not already that shape
    int64_t ldY64 = 0;                     // row pitch of Yd (ceil64(N): K1 loads without tests)

    // K1
    bool host_grad = false;                // pmx_set_host_grad: the gradient is whatever the caller uploaded into PMX_BUF_GA / GST (user `grad` callable)
    float* Tg[2][PMX_MAX_G] = {};           // bsdmm with a user-defined proxs_g member: its argument X + U_i, then its result (host round trip)
    float* Tf[2] = {};                      // bsdmm with a user-defined prox_f: its argument, then its result
    unsigned long long* k1prof = nullptr;  // tuning: phase cycle sums (PMX_K1_PROF=1)
    bool use_small = false;                // small problem (K <= 16, few million entries): k_grad_small in every mode
    bool use_bf16 = false;                 // split-bf16 kernel (mode BF16X3 / F16X2 and K <= 64), else exact fp32 MFMA
    bool use_f16 = false;                  // mode F16X2 at a shape the two-term fp16 kernel takes
    bool k128 = false;                     // mode F16X2, K = 128 at a shape k_grad_f16_k128 takes
    bool k32f16 = false;                   // mode F16X2, K = 32, M % 128 = 0, N % 256 = 0: k_grad_f16_k32 (no weights)
    bool f32pc = false;                    // exact-fp32 arithmetic at a shape the producer / consumer kernel k_grad_f32_pc takes
    bool f16_fell_back = false;            // use_f16, but the last launch ran the split-bf16 kernel (Y / W not fetchable in 8-byte pairs)
    bool f16_scales = false;               // use_f16 || k128: the K1 kernel needs the factor maxima (absmax) and max|Y|
    // [r4] the two-term fp16 kernels refuse a launch whose residual bound K max|A| max|S| exceeds rangeRatio max|Y| (f16_range_fault,
    // k_grad_f16_v8.hip): the context then continues in exact fp32 on the same frame (k1_leave_f16).  PMX_F16_RANGE=n: ratio 2^n, 0: no check
    float rangeRatio = 65536.f;
    int rangeFaults = 0;
    int f16_r3 = 0;                        // 1: k_grad_f16_v8<.., R3>, the residual with three terms per operand and two accumulators; [r5] 2: <.., HH>, the residual from the
                                           // high x high product and the rest as a correction slab (k_gfix.hip) where a kernel has that instance (K1's K = 64, 128; no weights), R3 elsewhere (PMX_F16_R3=0/1/2)
    float* fixPart = nullptr;              // k_gfix_gram's partial matrices [2][GFIX_PARTS][2][Kk * Kk]
    float* fixQ = nullptr;                 // [2][2][Kk * Kk]
    float* fixSlab[2] = {nullptr, nullptr};   // the correction slabs (rowsK[j] x Kk)
    long long* fixProf = nullptr;          // PMX_GFIX_PROF=1: time stamps of the correction kernels' first workgroup (printed by pmx_time_grad)
    bool fix_on = false;                   // the last gradient pass ran an <HH> kernel: the update kernels fold fixSlab behind K1's slabs (slab_ref)
    bool k1_sync_check = false;            // one-iteration-per-call paths (nothing to repeat into): every fp16 K1 launch is awaited and, refused, repeated in fp32 on the spot (enqueue_grad)
    int ncu = 0;
    _Float16* A16[2] = {nullptr, nullptr}; // k128: high / low fp16 terms of the scaled A (k_split_a_f16, once per K1 launch)
    float* absmax = nullptr;               // [3][256] partial maxima: |A|, |St| (per K1 launch), |Y| (at set_Y)
    float ymax = 0.f;                      // max |Y|
    float wmax = 1.f;                      // max(1, max |W|)
    // row-sharded adaprox: the last kernels enqueued were an iteration tail (k_ada_finish left the factor maxima in
    // `absmax`); every entry point that can change the factors otherwise clears it
    bool absmax_by_finish = false;
    // pgm: the partial Gram matrices in gramPart were left by the last k_pgm_update for the point the next iteration evaluates
    // (PgmArgs::gramPart): the step rule skips k_gram_partial.  Cleared wherever the factors can change behind the solver's back.
    bool gram_by_update = false;
    bool bsd_decide_pending = false;       // bsdmm: the Boyd test of the block updated last has not been enqueued yet (it rides in the next k_gram_reduce launch)
    BsdmmDecideArgs bsd_decide{};
    bool gram_fresh[2] = {false, false};   // bsdmm: gramPart[f] holds the partial Gram matrices of the CURRENT factor f (left by k_bsdmm_update)
    bool gram_in_update = true;            // PMX_GRAM_IN_UPDATE=0 (read at context creation): off
    bool fold_in_k1 = true;                // [r6] pgm: the step rule's Gram fold rides in K1's first workgroups where it can (PMX_FOLD_IN_K1=0: off, A/B)
    K1GramFold k1_fold{};                  //   what the NEXT K1 launch carries (part == nullptr: nothing); set by pgm_enqueue_iteration, consumed by enqueue_grad_once
    bool decide_pending = false;           // pgm: the stopping test of the last enqueued iteration has not been enqueued yet (it rides in the
                                           // next k_gram_reduce launch, or pgm_flush_decide() launches it at the end of a chunk)
    __bf16* Bp[2] = {nullptr, nullptr};    // presplit terms, row-major   [3][rowsPad][KP]
    __bf16* Bt[2] = {nullptr, nullptr};    // presplit terms, transposed  [2][KP][rowsPad]
    int64_t rowsPad[2] = {0, 0};
    GradPlan plan{};
    int chainL = 0;                        // > 0: the fp16 K1 sums gA in place along chains of this many workgroups (k_grad_f16_v8<CHAIN>)
    unsigned* chainFlags = nullptr;        // their arrival words
    unsigned chainSeq = 0;                 // launches so far (arrival words are monotonic: launch n counts from 64 n)
    int nSlabA = 0;                        // gA slabs the update kernels fold (plan.nSlabA, or one per chain group)
    int nSlabS = 0;                        // gSt slabs
    int chainFaults = 0;                   // times the chained mode was left after a fault
    // test hooks, read from the environment ONCE when the context is created (never on a launch path):
    int hook_inject_k1 = 0;                //   PMX_INJECT_K1_FAULT=n: the n-th chained K1 launch reports a fault
    int k1_prio = -999;                    // [r6] s_setprio level of K1's consumer waves (k1_set_priority); -999: the kernel's own default (k1_prio_for), PMX_K1_PRIO overrides
    std::string hook_tail_lockfile;        //   PMX_TAIL_LOCKFILE: several processes on ONE GPU take turns with the persistent tail
    bool tail_fused = false;               // adaprox: the iteration tail runs as one persistent kernel (k_ada_tail)
    GridBar* gridbar = nullptr;            // its barrier state
    long long* tailprof = nullptr;         // PMX_TAIL_PROF=1: phase time stamps of the last fused tail
    unsigned* tickets = nullptr;           // pgm: arrival counter of the update kernel's last-workgroup stopping test
    unsigned ticketLaunches = 0;           // tickets drawn by the update launches so far
    int tailFaults = 0;
    float* slab[2] = {nullptr, nullptr};
    const float* W = nullptr;              // weights of the likelihood (nullptr: W == 1), nmf.py:13-41
    int64_t ldW = 0;
    float* Wown = nullptr;
    double* lossPart = nullptr;
    int nloss = 0;                         // loss partials written by the last gradient launch

    // reductions / control
    double* partials = nullptr;            // [SL_COUNT][2][EW_BLOCKS]
    double* colpart = nullptr;             // [2][EW_BLOCKS][MAXK]
    float* gramPart = nullptr;             // [2][GRAM_BLOCKS][KP*KP]
    double* gramG = nullptr;               // [2][KP*KP]
    double* eigQ = nullptr;                // [2][KP*KP] Lanczos scratch
    DevStatus* dstatus = nullptr;
    DevStatus* hstatus = nullptr;          // pinned mirror

    // solver state
    Algo algo = ALG_NONE;
    pmx_pgm_params pgm{};
    pmx_adaprox_params ada{};
    pmx_bsdmm_params bsd{};
    int host_tau[2] = {0, 0};              // adaprox: passes of the host-side proximal loops of the current iteration
    int it = 0;                            // iterations enqueued AND completed (host view)
    double nest_t = 1.0;                   // NesterovAccelerator.t (utils.py:195)
    double btT[2] = {1.0, 1.0};            // backtracking step multipliers T (algorithms.py:85), never reset inside a run
    double bt_fprev = 0.0, bt_fnow = 0.0;
    bool bt_grad_fresh = false;            // line search: pmx_pgm_split(0) has just left this evaluation point's gradient in the slabs (a user `step` that takes
                                           // `grads`): the pmx_pgm_bt_split(0) that follows does not run K1 again.  Cleared by anything that moves the factors.
    int bt_pending = 0;                    // line search: blocks (bit j) whose user prox the caller owes (pmx_pgm_bt_split)
    int bt_trial = 3;                      // blocks whose sums the current trial renews
    float* btBuf[2] = {nullptr, nullptr};  // argument / result of a user prox inside the line search (PMX_BUF_BT_A / _ST)
    float omega_cur = 0.f;
    int nsub_guess = 2;
    int sub_nt = SUB_NT_MAX;               // proximal sub-iteration passes per launch (PMX_SUB_BATCH=1: one launch per pass)
    // row-sharded protocol: per enqueued iteration (index & 63: far more than a caller keeps in flight) the launch
    // size and the number of passes enqueued so far -- pmx_adaprox_more_subs continues the iteration the chain halted in
    struct SubRec { int it = -1, nt = 0, enq = 0; } sub_rec[64];
    std::vector<void*> allocs;

    // K1 timing (HIP events on the launch stream)
    bool timing = false;
    int timing_stride = 1;                 // bracket every n-th launch (the two event records cost ~9 us of stream time)
    unsigned timing_seq = 0;
    std::vector<hipEvent_t> ev;            // pairs
    size_t ev_used = 0;

    // per-phase timeline of the row-sharded iteration (pmx_set_phase_timing): events + their marks (0..6), in record order
    int ph_every = 0;
    unsigned ph_seq = 0;
    bool ph_cur = false, ph_prev = false;   // this / the previous iteration is a timed one
    std::vector<hipEvent_t> ph_ev;
    std::vector<int> ph_mark;

    // multi-GPU
    int rank = 0, world = 1;
    int64_t M_global = 0;
    float* comm = nullptr;                 // caller-owned all-reduce buffer
    bool comm_gram_dirty = true;           // S-split: the chunks' (unused) Gram sections have not been zeroed in this buffer yet
    bool shard_grad_from_comm = false;
    // S-split (pmx_set_s_split): the UPDATE of S is sharded too -- this rank owns the sncol columns of S from scol0 (rows of
    // S^T), their moments and proximal loop; gS arrives by reduce-scatter into comm_out, the updated columns leave by all-gather
    bool ssplit = false;
    int64_t scol0 = 0, sncol = 0;
    float* comm_out = nullptr;             // caller-owned: this rank's chunk after the reduce-scatter
    float* stepArr[2] = {nullptr, nullptr};   // pgm, split iterations: per-element steps of a user `step` that returned arrays (PMX_BUF_STEP_*)
    int step_arr_mask = 0;                   // bit j: block j's step comes from stepArr[j] in the split phases that follow
    void* rccl_comm = nullptr;             // pmx_comm_init: an RCCL communicator of this context's device (collectives run on its stream)
    int rccl_rank = 0, rccl_world = 0;
};

static int dalloc(pmx_ctx* c, void** p, size_t bytes, bool zero = true) {
    if (*p) return PMX_OK;
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) FAIL(PMX_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    c->allocs.push_back(*p);
    if (zero) HIP_CHECK(hipMemsetAsync(*p, 0, bytes ? bytes : 16, c->stream));
    return PMX_OK;
}
template <class T>
static int dallocT(pmx_ctx* c, T** p, size_t count, bool zero = true) { return dalloc(c, (void**)p, count * sizeof(T), zero); }

static ProxSeq to_dev(const pmx_proxseq& p) {
    ProxSeq d{};
    d.n = p.n;
    d.repeat = p.repeat < 1 ? 1 : p.repeat;
    for (int i = 0; i < PMX_MAX_SEQ; ++i) d.seq[i] = p.seq[i];
    return d;
}
// standalone: the stand-alone operator entry points (pmx_prox_apply / pmx_prox_array), which also normalise along the rows
static int check_prox(const pmx_proxseq& p, const char* what, bool standalone = false) {
    if (p.n < 0 || p.n > PMX_MAX_SEQ) FAIL(PMX_E_INVALID, "%s: bad operator count %d", what, p.n);
    for (int i = 0; i < p.n; ++i) {
        const pmx_prox& q = p.seq[i];
        if (q.op < PMX_PROX_ID || q.op > PMX_PROX_SOFT_PLUS) FAIL(PMX_E_INVALID, "%s: unknown prox op %d", what, q.op);
        if ((q.op == PMX_PROX_UNITY || q.op == PMX_PROX_UNITY_PLUS) && q.unit != 0 && !standalone)
            FAIL(PMX_E_UNSUPPORTED, "%s: prox_unity along the row dimension (numpy axis=0 on A / axis=1 on S) needs a grid-wide sum per "
                                    "application and is not part of the fused solver kernels (the host wrappers apply it between kernel "
                                    "launches through pmx_prox_apply)", what);
    }
    return PMX_OK;
}

// one operator sequence on a rows x K device array, incl. prox_unity* along the rows (column sums over all rows).
// `colpart` is scratch of EW_BLOCKS * MAXK doubles.
static int apply_prox_standalone(float* X, int64_t rows, int K, const pmx_proxseq& prox, const float* step_k, double* colpart, hipStream_t stream) {
    bool along_rows = false;
    for (int i = 0; i < prox.n; ++i)
        along_rows |= (prox.seq[i].op == PMX_PROX_UNITY || prox.seq[i].op == PMX_PROX_UNITY_PLUS) && prox.seq[i].unit != 0;
    ProxArgs a{};
    a.X = X;
    a.rows = rows;
    a.K = K;
    for (int k = 0; k < K; ++k) a.stepk[k] = step_k[k];
    if (!along_rows) {
        a.prox = to_dev(prox);
        launch_prox_apply(a, stream);
        return PMX_OK;
    }
    const int repeat = prox.repeat < 1 ? 1 : prox.repeat;
    for (int r = 0; r < repeat; ++r)
        for (int i = 0; i < prox.n; ++i) {
            pmx_proxseq one{};
            one.n = 1; one.repeat = 1; one.seq[0] = prox.seq[i];
            const bool rows_unity = (one.seq[0].op == PMX_PROX_UNITY || one.seq[0].op == PMX_PROX_UNITY_PLUS) && one.seq[0].unit != 0;
            if (!rows_unity) {
                a.prox = to_dev(one);
                launch_prox_apply(a, stream);
                continue;
            }
            if (one.seq[0].op == PMX_PROX_UNITY_PLUS) {           // plus first (operators.py:48-52)
                one.seq[0].op = PMX_PROX_PLUS; one.seq[0].unit = 0;
                a.prox = to_dev(one);
                launch_prox_apply(a, stream);
            }
            ColsumArgs cs{};
            cs.X[0] = X; cs.X[1] = X;
            cs.rows[0] = rows; cs.rows[1] = 0;
            cs.K = K;
            cs.colpart = colpart;
            cs.status = nullptr;
            launch_colsum(cs, stream);
            ColScaleArgs sc{};
            sc.X = X; sc.rows = rows; sc.K = K; sc.colpart = colpart;
            launch_colscale(sc, stream);
        }
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int pmx_abi_version(void) { return PMX_ABI_VERSION; }
extern "C" const char* pmx_last_error(void) { return g_err; }
extern "C" int pmx_abi_sizes(int sizes[5]) {
    if (!sizes) FAIL(PMX_E_INVALID, "NULL argument");
    sizes[0] = (int)sizeof(pmx_proxseq);
    sizes[1] = (int)sizeof(pmx_pgm_params);
    sizes[2] = (int)sizeof(pmx_adaprox_params);
    sizes[3] = (int)sizeof(pmx_bsdmm_params);
    sizes[4] = (int)sizeof(pmx_result);
    return PMX_OK;
}
extern "C" int pmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The producer / consumer K1s take M % 128 = 0 and N % 256 = 0 (128 at K = 128) only; the guarded kernels that take anything are
// 1.5-3 x slower.  A ragged problem with one of those K therefore runs on a zero-padded frame when that costs at most a quarter
// more entries: zero rows of A / St give zero rows of P, the padding of Y (and W) is zero, so R and both gradients are zero there
// and the loss is unchanged -- the arithmetic on the real entries is the aligned problem's.  PMX_FRAME=0 switches it off (A/B, tests).
static void choose_frame(int mode, int64_t M, int64_t N, int64_t K, int64_t* Mk, int64_t* Nk) {
    *Mk = M; *Nk = N;
    if (getenv("PMX_FRAME") && atoi(getenv("PMX_FRAME")) == 0) return;
    if (mode == PMX_MODE_F64 || grad_small_applies(M, N, K)) return;
    int64_t an = 256;
    if (K == 128) { if (mode != PMX_MODE_F16X2) return; an = 128; }          // k_grad_f16_k128
    else if (K == 32) { if (mode == PMX_MODE_BF16X3) return; }               // k_grad_f16_k32 / k_grad_f32_pc<32>
    else if (K != 64) return;                                                // k_grad_f16_v8 / k_grad_bf16_v7 / k_grad_f32_pc<64>
    const int64_t m = (M + 127) / 128 * 128, n = (N + an - 1) / an * an;
    if ((m == M && n == N) || (double)m * (double)n > 1.25 * (double)M * (double)N) return;
    *Mk = m; *Nk = n;
}

// which K1 runs on the frame Mk x Nk, its grid, and whether gA is summed along chains there
static void select_k1(pmx_ctx* c, int64_t Mk, int64_t Nk, int ncu) {
    const int64_t M = c->M, N = c->N, K = c->Kk;
    const int mode = c->mode;
    c->use_small = grad_small_applies(M, N, c->K);
    c->use_bf16 = (mode == PMX_MODE_BF16X3 || mode == PMX_MODE_F16X2) && K <= 64 && !c->use_small;
    c->plan = c->use_small ? grad_plan_small(M, N, K) : (c->use_bf16 ? grad_plan_bf16(Mk, Nk, K) : grad_plan_f32(Mk, Nk, K));
    c->use_f16 = mode == PMX_MODE_F16X2 && c->use_bf16 && grad_bf16_takes_weights(c->plan, Mk, Nk, K);   // same shapes as v7
    c->k128 = mode == PMX_MODE_F16X2 && !c->use_small && grad_k128_applies(Mk, Nk, K);
    if (c->k128) c->plan = grad_plan_k128(Mk, Nk);
    c->k32f16 = mode == PMX_MODE_F16X2 && !c->use_small && grad_f16_k32_applies(Mk, Nk, K);
    if (c->k32f16) { c->plan = grad_plan_f16_k32(Mk, Nk); c->use_bf16 = false; c->use_f16 = false; }
    c->f32pc = !c->use_small && !c->use_bf16 && !c->k128 && !c->k32f16 && grad_f32pc_applies(Mk, Nk, K);
    if (c->f32pc) c->plan = grad_plan_f32pc(Mk, Nk, K);
    c->f16_scales = c->use_f16 || c->k128 || c->k32f16;
    c->nSlabA = c->plan.nSlabA;
    c->nSlabS = c->plan.nSlabS;
    c->chainL = 0;
    const bool v7_shape = c->use_bf16 && grad_bf16_takes_weights(c->plan, Mk, Nk, K);     // k_grad_bf16_v7 / k_grad_f16_v8: the chained frame
    if (v7_shape || c->f32pc || c->k128) {
        c->chainL = grad_chain_length(c->plan, Mk, ncu, (c->use_f16 || c->k128) ? 16 : 32);
        if (c->chainL > 0) c->nSlabA = c->plan.gridY / c->chainL;
    }
}

extern "C" int pmx_ctx_create(pmx_ctx** out, int device, int64_t M, int64_t N, int64_t K, int mode, void* stream) {
    if (!out) FAIL(PMX_E_INVALID, "out is NULL");
    if (M <= 0 || N <= 0 || K <= 0) FAIL(PMX_E_INVALID, "bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    if (K > MAXK) FAIL(PMX_E_UNSUPPORTED, "K=%lld > %d components is not supported", (long long)K, MAXK);
    if (M > (1ll << 30) || N > (1ll << 30)) FAIL(PMX_E_UNSUPPORTED, "dimension too large");
    const bool f64_mfma = mode == PMX_MODE_F64_MFMA;
    if (f64_mfma) mode = PMX_MODE_F64;           // the matrix-core kernels whatever the shape
    const bool want_r3 = mode == PMX_MODE_F16X2R;
    if (want_r3) mode = PMX_MODE_F16X2;          // the same kernels, frames and fall-backs; k_grad_f16_v8 runs its <R3> instance
    if (mode != PMX_MODE_F32 && mode != PMX_MODE_BF16X3 && mode != PMX_MODE_F16X2 && mode != PMX_MODE_F64) FAIL(PMX_E_UNSUPPORTED, "compute mode %d is not built into this library", mode);
    const bool f64_small = !f64_mfma && grad_small_applies(M, N, K) && K <= 16 && M <= 8192 && N <= 8192;
    const bool f64_big_off = getenv("PMX_F64_BIG") && atoi(getenv("PMX_F64_BIG")) == 0;      // (A/B, and the way back to the fp32 computation of large fp64 problems)
    if (mode == PMX_MODE_F64 && !f64_small && (f64_big_off || (double)M * (double)N * 8.0 > 160e9))
        FAIL(PMX_E_UNSUPPORTED, "fp64 arithmetic is not available for %lld x %lld x %lld (%s); it runs in fp32",
             (long long)M, (long long)N, (long long)K, f64_big_off ? "PMX_F64_BIG=0: small problems only" : "Y does not fit");
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) FAIL(PMX_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_CHECK(hipSetDevice(device));
    pmx_ctx* c = new pmx_ctx();
    c->device = device;
    c->M = M; c->N = N; c->K = K;
    c->rows[0] = M; c->rows[1] = N;
    c->M_global = M;
    c->mode = mode;
    c->KP = K <= 32 ? 32 : (K <= 64 ? 64 : 128);
    if (stream) c->stream = (hipStream_t)stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; FAIL(PMX_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    if (const char* e = getenv("PMX_GRAM_IN_UPDATE")) c->gram_in_update = atoi(e) != 0;
    if (const char* e = getenv("PMX_INJECT_K1_FAULT")) c->hook_inject_k1 = atoi(e);
    if (const char* e = getenv("PMX_FOLD_IN_K1")) c->fold_in_k1 = atoi(e) != 0;
    if (const char* e = getenv("PMX_K1_PRIO")) c->k1_prio = atoi(e);
    if (const char* e = getenv("PMX_TAIL_LOCKFILE")) c->hook_tail_lockfile = e;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ncu = 0;
    c->ncu = ncu;
    c->f16_r3 = want_r3 ? 2 : 0;
    if (const char* e = getenv("PMX_F16_R3")) c->f16_r3 = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));        // (A/B switch: any f16x2 context)
    if (const char* e = getenv("PMX_F16_RANGE")) c->rangeRatio = atoi(e) > 0 ? ldexpf(1.f, atoi(e)) : 0.f;
    c->Kk = K;
    if (!(getenv("PMX_FRAME") && atoi(getenv("PMX_FRAME")) == 0) && mode != PMX_MODE_F64 && !grad_small_applies(M, N, K)) {
        if (K < 32 && mode != PMX_MODE_BF16X3) c->Kk = 32;           // k_grad_f16_k32 / k_grad_f32_pc<32>
        else if (K > 32 && K < 64) c->Kk = 64;                       // k_grad_f16_v8 / k_grad_bf16_v7 / k_grad_f32_pc<64>
        else if (K > 64 && K < 128 && mode == PMX_MODE_F16X2) c->Kk = 128;   // k_grad_f16_k128
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        choose_frame(mode, M, N, c->Kk, &c->Mk, &c->Nk);
        c->framed = c->Mk != M || c->Nk != N;
        select_k1(c, c->Mk, c->Nk, ncu);
        const bool tuned = c->k128 || c->k32f16 || c->f32pc || (c->use_bf16 && grad_bf16_takes_weights(c->plan, c->Mk, c->Nk, c->Kk));
        if (c->Kk == K || tuned) break;
        c->Kk = K;                                                   // (padding K buys nothing where the shape keeps the guarded kernels)
    }
    if (c->framed && c->chainL == 0 && (c->use_bf16 || c->f32pc || c->k128)) {
        // a frame the chained accumulation of gA does not take (e.g. 125 panels x 63 column regions): a few more panels / column
        // regions often make one that it does (128 x 64) -- worth up to 6 % more entries (the chain is ~10 % of an iteration: one
        // gA slab per 16 column regions instead of one each, for K1 to write and the update kernel to fold)
        const int64_t an = c->Kk == 128 ? 128 : 256, m0 = c->Mk, n0 = c->Nk;
        int64_t bm = 0, bn = 0;
        for (int i = 0; i <= 8; ++i)
            for (int j = 0; j <= 8; ++j) {
                const int64_t m = m0 + 128 * i, n = n0 + an * j;
                if ((double)m * (double)n > 1.06 * (double)m0 * (double)n0 || (double)m * (double)n > 1.25 * (double)M * (double)N) continue;
                if (bm && m * n >= bm * bn) continue;
                select_k1(c, m, n, ncu);
                if (c->chainL > 0) { bm = m; bn = n; }
            }
        if (bm) { c->Mk = bm; c->Nk = bn; }
        select_k1(c, c->Mk, c->Nk, ncu);
    }
    c->rowsK[0] = c->Mk; c->rowsK[1] = c->Nk;
    const int64_t Mk = c->Mk, Nk = c->Nk;
    if (mode == PMX_MODE_F64) {              // fp64 context: its own arrays and kernels (k_small_f64.hip), nothing of the fp32 state
        c->f64 = true;
        c->f64big = !f64_small;
        c->use_small = f64_small;
        c->use_bf16 = c->use_f16 = c->k128 = c->f32pc = c->f16_scales = false;
        c->chainL = 0;
        if (c->f64big) {                     // k_big_f64.hip: one MFMA pass per gradient, a few slabs each
            pass64_plan(M, N, (int)K, &c->nsplit64[0], &c->bps64[0]);
            pass64_plan(N, M, (int)K, &c->nsplit64[1], &c->bps64[1]);
            c->nSlabA = c->nsplit64[0]; c->nSlabS = c->nsplit64[1];
            c->t64x = (int)((M + 63) / 64) * c->nsplit64[0];       // workgroups of the pass (-> loss partials)
            c->t64y = (int)((N + 63) / 64) * c->nsplit64[1];
            c->plan.gridX = c->t64y; c->plan.gridY = 1; c->plan.RP = 1;
        } else {
            c->t64x = (int)((N + SG_COLS - 1) / SG_COLS);
            c->t64y = (int)((M + S64_ROWS - 1) / S64_ROWS);
            c->nSlabA = c->t64x; c->nSlabS = c->t64y;
            c->plan.gridX = c->t64y; c->plan.gridY = c->t64x; c->plan.RP = 1;
        }
        const int64_t Mp64 = (M + 63) / 64 * 64, Np64 = (N + 63) / 64 * 64;
        c->ldY64 = c->f64big ? Np64 : N;
        // (the matrix-core path allocates Y when it arrives: a context that only evaluates the step rule -- nmf.step_pgm on fp64 factors -- holds nothing M x N)
        int rc64 = c->f64big ? PMX_OK : dallocT(c, &c->Yd, (size_t)M * N, false);
        for (int j = 0; j < 2 && rc64 == PMX_OK && c->f64big; ++j)
            if (pad64_needed(c->rows[j], (int)K, c->KP)) rc64 = dallocT(c, &c->Xk64[j], (size_t)((c->rows[j] + 63) / 64 * 64) * c->KP);
        for (int j = 0; j < 2 && rc64 == PMX_OK; ++j) {
            rc64 = dallocT(c, &c->Xd[j], (size_t)c->rows[j] * K);
            if (rc64 == PMX_OK) rc64 = dallocT(c, &c->Gd[j], (size_t)c->rows[j] * K);
        }
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->slabd[0], (size_t)c->nSlabA * M * K, false);
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->slabd[1], (size_t)c->nSlabS * N * K, false);
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->lossPart, c->f64big ? (size_t)std::max(c->t64x, c->t64y) : (size_t)c->t64x * c->t64y);
        if (rc64 == PMX_OK && c->f64big) rc64 = dallocT(c, &c->gramPart64, (size_t)2 * G64_BLOCKS * c->KP * c->KP, false);
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->partials, (size_t)SL_COUNT * 2 * EW_BLOCKS);
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->gramG, (size_t)2 * c->KP * c->KP);
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->eigQ, (size_t)2 * c->KP * c->KP);
        if (rc64 == PMX_OK) rc64 = dallocT(c, &c->dstatus, 1);
        if (rc64 == PMX_OK) {
            hipError_t e = hipHostMalloc((void**)&c->hstatus, sizeof(DevStatus), hipHostMallocDefault);
            if (e != hipSuccess) { pmx_set_error("hipHostMalloc: %s", hipGetErrorString(e)); rc64 = PMX_E_NOMEM; }
        }
        if (rc64 != PMX_OK) { pmx_ctx_destroy(c); return rc64; }
        *out = c;
        return PMX_OK;
    }
    int rc = PMX_OK;
    if (c->chainL > 0) rc = dallocT(c, &c->chainFlags, (size_t)(c->plan.gridX * c->plan.gridY / c->chainL) * c->plan.RP * 4);
    if (c->f16_scales) rc = dallocT(c, &c->absmax, (size_t)3 * 256);
    if (rc == PMX_OK && c->f16_r3 == 2 && (c->use_f16 || c->k128)) {          // <HH> instances: the correction's matrices and slabs
        rc = dallocT(c, &c->fixPart, (size_t)2 * GFIX_PARTS * 2 * c->Kk * c->Kk, false);
        if (rc == PMX_OK) rc = dallocT(c, &c->fixQ, (size_t)4 * c->Kk * c->Kk);
        for (int j = 0; j < 2 && rc == PMX_OK; ++j) rc = dallocT(c, &c->fixSlab[j], (size_t)c->rowsK[j] * c->Kk);
        if (rc == PMX_OK && getenv("PMX_GFIX_PROF")) rc = dallocT(c, &c->fixProf, 16);
    }
    for (int t = 0; t < 2 && rc == PMX_OK && c->k128; ++t) rc = dallocT(c, &c->A16[t], (size_t)Mk * c->Kk, c->framed || c->Kk != K);   // (framed: the rows behind M / the columns behind K stay zero)
    for (int j = 0; j < 2 && rc == PMX_OK && c->Kk != K; ++j) rc = dallocT(c, &c->Xk[j], (size_t)c->rowsK[j] * c->Kk);          // K1's zero-padded operands
    if (c->use_bf16) {
        for (int j = 0; j < 2 && rc == PMX_OK; ++j) {
            c->rowsPad[j] = (c->rowsK[j] + 127) / 128 * 128;
            rc = dallocT(c, &c->Bp[j], (size_t)3 * c->rowsPad[j] * c->KP, false);
            if (rc == PMX_OK) rc = dallocT(c, &c->Bt[j], (size_t)2 * c->KP * c->rowsPad[j], false);
        }
    }
    for (int j = 0; j < 2 && rc == PMX_OK; ++j) {
        rc = dallocT(c, &c->X[j], (size_t)c->rowsK[j] * K);
        if (rc == PMX_OK) rc = dallocT(c, &c->G[j], (size_t)c->rows[j] * K);
    }
    if (rc == PMX_OK) rc = dallocT(c, &c->slab[0], (size_t)c->nSlabA * Mk * c->Kk, false);
    if (rc == PMX_OK) rc = dallocT(c, &c->slab[1], (size_t)c->plan.nSlabS * Nk * c->Kk, false);
    if (rc == PMX_OK) rc = dallocT(c, &c->lossPart, (size_t)2 * c->plan.gridX * c->plan.gridY);
    if (rc == PMX_OK) rc = dallocT(c, &c->partials, (size_t)SL_COUNT * 2 * EW_BLOCKS);
    if (rc == PMX_OK) rc = dallocT(c, &c->colpart, (size_t)2 * EW_BLOCKS * MAXK);
    if (rc == PMX_OK) rc = dallocT(c, &c->gramPart, (size_t)2 * GRAM_BLOCKS * c->KP * c->KP);
    if (rc == PMX_OK) rc = dallocT(c, &c->gramG, (size_t)2 * c->KP * c->KP);
    if (rc == PMX_OK) rc = dallocT(c, &c->eigQ, (size_t)2 * c->KP * c->KP);
    if (rc == PMX_OK) rc = dallocT(c, &c->dstatus, 1);
    if (rc == PMX_OK) {
        hipError_t e = hipHostMalloc((void**)&c->hstatus, sizeof(DevStatus), hipHostMallocDefault);
        if (e != hipSuccess) { pmx_set_error("hipHostMalloc: %s", hipGetErrorString(e)); rc = PMX_E_NOMEM; }
    }
    if (rc != PMX_OK) { pmx_ctx_destroy(c); return rc; }
    *out = c;
    return PMX_OK;
}

extern "C" int pmx_ctx_destroy(pmx_ctx* c) {
    if (!c) return PMX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)pmx_comm_destroy(c);
    for (void* p : c->allocs) (void)hipFree(p);
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& e : c->ph_ev) (void)hipEventDestroy(e);
    if (c->hstatus) (void)hipHostFree(c->hstatus);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PMX_OK;
}

extern "C" int pmx_set_timing(pmx_ctx* c, int on) {
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (on && c->ev.empty()) {
        c->ev.resize(2 * 8192);
        for (auto& e : c->ev) HIP_CHECK(hipEventCreate(&e));
    }
    c->timing = on != 0;
    c->timing_stride = on > 1 ? on : 1;
    c->timing_seq = 0;
    c->ev_used = 0;
    return PMX_OK;
}

extern "C" int pmx_get_timing(pmx_ctx* c, double* total_ms, int* launches) {
    if (!c || !total_ms || !launches) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    // launches that found the chain halted are no-ops (~1 us): they are not K1 executions and are excluded
    std::vector<float> d;
    float mx = 0.f;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        d.push_back(ms);
        mx = std::max(mx, ms);
    }
    double tot = 0.0;
    int n = 0;
    for (float ms : d)
        if (ms >= 0.05f * mx) { tot += ms; ++n; }
    *total_ms = tot;
    *launches = n;
    return PMX_OK;
}

// mark m of the sharded iteration's timeline: 0 phase-0 start, 1 K1 done, 2 phase-0 end, 3 phase-1 start, 4 post done,
// 5 phase-1 end, 6 next phase-0 start (recorded for the iteration BEFORE the one mark 0 opens)
static int phase_mark(pmx_ctx* c, int m) {
    if (c->ph_every <= 0) return PMX_OK;
    if (m == 0) {
        if (c->ph_prev && c->ph_mark.size() < c->ph_ev.size()) {
            HIP_CHECK(hipEventRecord(c->ph_ev[c->ph_mark.size()], c->stream));
            c->ph_mark.push_back(6);
        }
        c->ph_cur = (c->ph_seq++ % (unsigned)c->ph_every) == 0 && c->ph_mark.size() + 8 <= c->ph_ev.size();
        c->ph_prev = c->ph_cur;
    }
    if (!c->ph_cur) return PMX_OK;
    HIP_CHECK(hipEventRecord(c->ph_ev[c->ph_mark.size()], c->stream));
    c->ph_mark.push_back(m);
    return PMX_OK;
}
extern "C" int pmx_set_phase_timing(pmx_ctx* c, int every) {
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (every > 0 && c->ph_ev.empty()) {
        c->ph_ev.resize(4096);
        for (auto& e : c->ph_ev) HIP_CHECK(hipEventCreate(&e));
    }
    c->ph_every = every > 0 ? every : 0;
    c->ph_seq = 0;
    c->ph_cur = c->ph_prev = false;
    c->ph_mark.clear();
    return PMX_OK;
}
extern "C" int pmx_get_phase_timing(pmx_ctx* c, double ms[6], int* iterations) {
    if (!c || !ms || !iterations) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    double sum[6] = {0, 0, 0, 0, 0, 0};
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i + 1 < c->ph_mark.size(); ++i) {
        const int a = c->ph_mark[i], b = c->ph_mark[i + 1];
        if (b != a + 1 || a > 5) continue;               // (a mark-6 event closes an iteration; 0 opens the next)
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, c->ph_ev[i], c->ph_ev[i + 1]));
        sum[a] += t;
        cnt[a] += 1;
    }
    for (int a = 0; a < 6; ++a) ms[a] = cnt[a] ? sum[a] / cnt[a] : 0.0;
    *iterations = cnt[0];
    return PMX_OK;
}

extern "C" int pmx_k1_info(pmx_ctx* c, int info[8]) {
    if (!c || !info) FAIL(PMX_E_INVALID, "NULL argument");
    info[0] = c->f64big ? 13 : c->f64 ? 7 : c->k32f16 ? (c->f16_r3 ? 10 : 8) : c->use_small ? 4 : (c->k128 ? (c->f16_r3 == 2 && !c->W && c->fixPart ? 12 : 5) : (c->use_f16 && !c->f16_fell_back ? (c->f16_r3 && !c->W ? (c->f16_r3 == 2 && c->fixPart ? 11 : 9) : 2) : (c->use_bf16 ? 1 : (c->f32pc ? 6 : 0))));
    info[1] = c->chainL;
    info[2] = c->nSlabA;
    info[3] = c->nSlabS;
    info[4] = c->plan.gridX;
    info[5] = c->plan.gridY;
    info[6] = c->plan.RP;
    info[7] = c->chainFaults + 1000 * c->tailFaults + (c->tail_fused ? 1000000 : 0) + 10000000 * std::min(c->rangeFaults, 9);
    return PMX_OK;
}

extern "C" int pmx_k1_frame(pmx_ctx* c, int64_t frame[3]) {
    if (!c || !frame) FAIL(PMX_E_INVALID, "NULL argument");
    frame[0] = c->f64 ? c->M : c->Mk;
    frame[1] = c->f64 ? c->N : c->Nk;
    frame[2] = c->f64 ? c->K : c->Kk;
    return PMX_OK;
}

extern "C" int pmx_ctx_sync(pmx_ctx* c) {
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

// max |Y| for the fp16 path's residual scale (one pass over Y, once)
static int measure_ymax(pmx_ctx* c) {
    if (!c->f16_scales) return PMX_OK;
    launch_absmax_pitched(c->Y, c->ldY, c->M, c->N, c->absmax + 512, c->stream);
    HIP_CHECK(hipGetLastError());
    float h[256];
    HIP_CHECK(hipMemcpyAsync(h, c->absmax + 512, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    float m = 0.f;
    for (float v : h) m = v > m ? v : m;
    if (!(m < 3.0e38f)) FAIL(PMX_E_INVALID, "Y contains non-finite values");
    c->ymax = m;
    return PMX_OK;
}

// the context's own copy of an M x N array (Y, W) in K1's frame: Mk x Nk, zero outside M x N (the buffer is zeroed when it is created
// and nothing ever writes there)
static int own_frame_copy(pmx_ctx* c, float** own, const float* src, int64_t ld, hipMemcpyKind kind) {
    if (!*own) {
        int rc = dallocT(c, own, (size_t)c->Mk * c->Nk, c->framed);
        if (rc != PMX_OK) return rc;
    }
    HIP_CHECK(hipMemcpy2DAsync(*own, c->Nk * sizeof(float), src, ld * sizeof(float), c->N * sizeof(float), c->M, kind, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_set_Y_host(pmx_ctx* c, const float* Y, int64_t ld) {
    if (!c || !Y) FAIL(PMX_E_INVALID, "NULL argument");
    if (c->f64) FAIL(PMX_E_UNSUPPORTED, "an fp64 context takes Y through pmx_set_Y_host_f64");
    if (ld < c->N) FAIL(PMX_E_INVALID, "ld %lld < N", (long long)ld);
    HIP_CHECK(hipSetDevice(c->device));
    int rc = own_frame_copy(c, &c->Yown, Y, ld, hipMemcpyHostToDevice);
    if (rc != PMX_OK) return rc;
    c->Y = c->Yown;
    c->ldY = c->Nk;
    c->haveY = true;
    return measure_ymax(c);
}

extern "C" int pmx_set_Y_device(pmx_ctx* c, const float* dY, int64_t ld, int copy) {
    if (!c || !dY) FAIL(PMX_E_INVALID, "NULL argument");
    if (c->f64) FAIL(PMX_E_UNSUPPORTED, "an fp64 context takes Y through pmx_set_Y_host_f64");
    if (ld < c->N) FAIL(PMX_E_INVALID, "ld %lld < N", (long long)ld);
    HIP_CHECK(hipSetDevice(c->device));
    if (copy || c->framed) {                 // (a framed context always works on its own zero-padded copy: pmx.h)
        int rc = own_frame_copy(c, &c->Yown, dY, ld, hipMemcpyDeviceToDevice);
        if (rc != PMX_OK) return rc;
        c->Y = c->Yown;
        c->ldY = c->Nk;
    } else {
        c->Y = dY;
        c->ldY = ld;
    }
    c->haveY = true;
    return measure_ymax(c);
}

// max(1, max |W|) for the fp16 path's residual scale
static int measure_wmax(pmx_ctx* c) {
    c->wmax = 1.f;
    if (!c->f16_scales || !c->W) return PMX_OK;      // (the two-term fp16 kernels, K = 64 and K = 128: max |W| enters R's scale)
    launch_absmax_pitched(c->W, c->ldW, c->M, c->N, c->absmax + 512, c->stream);
    HIP_CHECK(hipGetLastError());
    float h[256];
    HIP_CHECK(hipMemcpyAsync(h, c->absmax + 512, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    float m = 1.f;
    for (float v : h) m = v > m ? v : m;
    if (!(m < 3.0e38f)) FAIL(PMX_E_INVALID, "W contains non-finite values");
    c->wmax = m;
    return PMX_OK;
}

static int set_W_common(pmx_ctx* c, const float* W, int64_t ld, int from_host, int copy) {
    if (c && c->f64 && W) FAIL(PMX_E_UNSUPPORTED, "weights are not implemented in fp64 contexts");
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (!W) { c->W = nullptr; c->ldW = 0; c->wmax = 1.f; return PMX_OK; }
    if (ld < c->N) FAIL(PMX_E_INVALID, "ld %lld < N", (long long)ld);
    if (c->k32f16) FAIL(PMX_E_UNSUPPORTED, "k_grad_f16_k32 takes no weights; create the context with PMX_MODE_F32");
    if (c->use_bf16 && !grad_bf16_takes_weights(c->plan, c->Mk, c->Nk, c->Kk))
        FAIL(PMX_E_UNSUPPORTED, "a weighted likelihood in a split-precision mode needs K = 64 with M %% 128 = 0, N %% 256 = 0 or K = 128 with M %% 128 = 0, N %% 128 = 0; create the context with PMX_MODE_F32");
    if (c->comm) FAIL(PMX_E_UNSUPPORTED, "weights are not supported in row-sharded runs");
    HIP_CHECK(hipSetDevice(c->device));
    if (from_host || copy || c->framed) {
        int rc = own_frame_copy(c, &c->Wown, W, ld, from_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
        if (rc != PMX_OK) return rc;
        c->W = c->Wown;
        c->ldW = c->Nk;
    } else {
        c->W = W;
        c->ldW = ld;
    }
    return measure_wmax(c);
}
extern "C" int pmx_set_W_host(pmx_ctx* c, const float* W, int64_t ld) { return set_W_common(c, W, ld, 1, 1); }
extern "C" int pmx_set_W_device(pmx_ctx* c, const float* dW, int64_t ld, int copy) { return set_W_common(c, dW, ld, 0, copy); }

static int buf_lookup(pmx_ctx* c, int buf, float*** slot, int64_t* count, bool create) {
    if (c->f64) FAIL(PMX_E_UNSUPPORTED, "an fp64 context has no float arrays: pmx_upload_f64 / pmx_download_f64");
    int j;
    float** p = nullptr;
    if (buf >= PMX_BUF_A && buf <= PMX_BUF_PSI_ST) {
        j = buf & 1;
        switch (buf >> 1) {
            case 0: p = &c->X[j]; break;
            case 1: p = &c->G[j]; break;
            case 2: p = &c->Mm[j]; break;
            case 3: p = &c->Vv[j]; break;
            case 4: p = &c->Vh[j]; break;
            case 5: p = (c->algo == ALG_PGM && c->pgm.accelerated) ? &c->Xe[j] : &c->X[j]; break;
            case 6: p = &c->Xp[j]; break;
            case 7: p = &c->Psi[j]; break;
        }
    } else if (buf >= PMX_BUF_Z0 && buf < PMX_BUF_Z0 + 2 * PMX_MAX_G) {
        j = (buf - PMX_BUF_Z0) / PMX_MAX_G;
        p = &c->Zg[j][(buf - PMX_BUF_Z0) % PMX_MAX_G];
    } else if (buf >= PMX_BUF_U0 && buf < PMX_BUF_U0 + 2 * PMX_MAX_G) {
        j = (buf - PMX_BUF_U0) / PMX_MAX_G;
        p = &c->Ug[j][(buf - PMX_BUF_U0) % PMX_MAX_G];
    } else if (buf >= PMX_BUF_TG0 && buf < PMX_BUF_TG0 + 2 * PMX_MAX_G) {
        j = (buf - PMX_BUF_TG0) / PMX_MAX_G;
        p = &c->Tg[j][(buf - PMX_BUF_TG0) % PMX_MAX_G];
    } else if (buf == PMX_BUF_STEP_A || buf == PMX_BUF_STEP_ST) {
        j = buf - PMX_BUF_STEP_A;
        p = &c->stepArr[j];
    } else if (buf == PMX_BUF_BT_A || buf == PMX_BUF_BT_ST) {
        j = buf - PMX_BUF_BT_A;
        p = &c->btBuf[j];
    } else FAIL(PMX_E_INVALID, "unknown buffer id %d", buf);
    *count = c->rows[j] * c->K;
    if (!*p) {
        if (!create) FAIL(PMX_E_STATE, "buffer %d has not been created yet", buf);
        int rc = dallocT(c, p, (size_t)c->rowsK[j] * c->K);     // (K1 may read it: the frame's rows, zero behind the real ones)
        if (rc != PMX_OK) return rc;
    }
    *slot = p;
    return PMX_OK;
}

extern "C" int pmx_upload(pmx_ctx* c, int buf, const float* host, int64_t count) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; c->bt_grad_fresh = false; }
    if (!c || !host) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    float** slot; int64_t n;
    int rc = buf_lookup(c, buf, &slot, &n, true);
    if (rc != PMX_OK) return rc;
    if (count != n) FAIL(PMX_E_INVALID, "buffer %d holds %lld floats, got %lld", buf, (long long)n, (long long)count);
    HIP_CHECK(hipMemcpyAsync(*slot, host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_download(pmx_ctx* c, int buf, float* host, int64_t count) {
    if (!c || !host) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    float** slot; int64_t n;
    int rc = buf_lookup(c, buf, &slot, &n, false);
    if (rc != PMX_OK) return rc;
    if (count != n) FAIL(PMX_E_INVALID, "buffer %d holds %lld floats, got %lld", buf, (long long)n, (long long)count);
    HIP_CHECK(hipMemcpyAsync(host, *slot, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_buffer_ptr(pmx_ctx* c, int buf, void** dptr, int64_t* count) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    if (!c || !dptr) FAIL(PMX_E_INVALID, "NULL argument");
    float** slot; int64_t n;
    int rc = buf_lookup(c, buf, &slot, &n, true);
    if (rc != PMX_OK) return rc;
    *dptr = *slot;
    if (count) *count = n;
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
// PMX_MODE_F64: transfers of an fp64 context (k_small_f64.hip)
// ------------------------------------------------------------------------------------------------
// the matrix-core path's Y: ceil64(M) x ceil64(N), zero behind the real extents (allocated on first use)
static int alloc_Y64(pmx_ctx* c) {
    if (c->Yd || !c->f64big) return PMX_OK;
    const int64_t Mp = (c->M + 63) / 64 * 64;
    return dallocT(c, &c->Yd, (size_t)Mp * c->ldY64, Mp != c->M || c->ldY64 != c->N);
}
extern "C" int pmx_set_Y_host_f64(pmx_ctx* c, const double* Y, int64_t ld) {
    if (!c || !Y) FAIL(PMX_E_INVALID, "NULL argument");
    if (!c->f64) FAIL(PMX_E_STATE, "pmx_set_Y_host_f64 needs a PMX_MODE_F64 context");
    if (ld < c->N) FAIL(PMX_E_INVALID, "ld %lld < N", (long long)ld);
    HIP_CHECK(hipSetDevice(c->device));
    int rcy = alloc_Y64(c);
    if (rcy != PMX_OK) return rcy;
    HIP_CHECK(hipMemcpy2DAsync(c->Yd, (c->f64big ? c->ldY64 : c->N) * sizeof(double), Y, ld * sizeof(double), c->N * sizeof(double), c->M, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->haveY = true;
    return PMX_OK;
}
extern "C" int pmx_set_Y_device_f64(pmx_ctx* c, const double* dY, int64_t ld) {
    if (!c || !dY) FAIL(PMX_E_INVALID, "NULL argument");
    if (!c->f64) FAIL(PMX_E_STATE, "pmx_set_Y_device_f64 needs a PMX_MODE_F64 context");
    if (ld < c->N) FAIL(PMX_E_INVALID, "ld %lld < N", (long long)ld);
    HIP_CHECK(hipSetDevice(c->device));
    int rcy = alloc_Y64(c);
    if (rcy != PMX_OK) return rcy;
    HIP_CHECK(hipMemcpy2DAsync(c->Yd, (c->f64big ? c->ldY64 : c->N) * sizeof(double), dY, ld * sizeof(double), c->N * sizeof(double), c->M, hipMemcpyDeviceToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->haveY = true;
    return PMX_OK;
}
extern "C" int pmx_set_W_host_f64(pmx_ctx* c, const double* W, int64_t ld) {
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (!c->f64) FAIL(PMX_E_STATE, "pmx_set_W_host_f64 needs a PMX_MODE_F64 context");
    if (!W) { c->Wd_on = false; return PMX_OK; }
    if (!c->f64big) FAIL(PMX_E_UNSUPPORTED, "the small-problem fp64 kernels take no weights; create the context with PMX_MODE_F64_MFMA");
    if (ld < c->N) FAIL(PMX_E_INVALID, "ld %lld < N", (long long)ld);
    HIP_CHECK(hipSetDevice(c->device));
    const int64_t Mp = (c->M + 63) / 64 * 64;
    int rc = dallocT(c, &c->Wd, (size_t)Mp * c->ldY64);       // (zeroed once: the padding stays zero)
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipMemcpy2DAsync(c->Wd, c->ldY64 * sizeof(double), W, ld * sizeof(double), c->N * sizeof(double), c->M, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->Wd_on = true;
    return PMX_OK;
}
static int buf_lookup_f64(pmx_ctx* c, int buf, bool writable, double** p, int64_t* count) {
    if (!c->f64) FAIL(PMX_E_STATE, "not a PMX_MODE_F64 context");
    const int j = buf & 1;
    switch (buf) {
        case PMX_BUF_A: case PMX_BUF_ST: *p = c->Xd[j]; break;
        case PMX_BUF_GA: case PMX_BUF_GST: if (writable) FAIL(PMX_E_INVALID, "the gradient buffers of an fp64 context are read-only"); *p = c->Gd[j]; break;
        case PMX_BUF_EVAL_A: case PMX_BUF_EVAL_ST: if (writable) FAIL(PMX_E_INVALID, "buffer %d is read-only", buf);
            *p = (c->algo == ALG_PGM && c->pgm.accelerated && c->Xed[j]) ? c->Xed[j] : c->Xd[j]; break;
        case PMX_BUF_MA: case PMX_BUF_MST: case PMX_BUF_VA: case PMX_BUF_VST: case PMX_BUF_VHA: case PMX_BUF_VHST: {
            double** slot = (buf == PMX_BUF_MA || buf == PMX_BUF_MST) ? &c->Md[j] : ((buf == PMX_BUF_VA || buf == PMX_BUF_VST) ? &c->Vd[j] : &c->Vhd[j]);
            if (!*slot) {
                if (!writable) FAIL(PMX_E_STATE, "buffer %d has not been created yet", buf);
                int rc = dallocT(c, slot, (size_t)c->rows[j] * c->K);
                if (rc != PMX_OK) return rc;
            }
            *p = *slot;
            break;
        }
        default:
            if (buf >= PMX_BUF_Z0 && buf < PMX_BUF_U0 + 2 * PMX_MAX_G) {       // bsdmm's constraint variables (read-only: tests compare them with the oracle's)
                const bool isU = buf >= PMX_BUF_U0;
                const int k = buf - (isU ? PMX_BUF_U0 : PMX_BUF_Z0), jj = k / PMX_MAX_G, i = k % PMX_MAX_G;
                double* q = isU ? c->Ud[jj][i] : c->Zd[jj][i];
                if (writable || !q) FAIL(PMX_E_STATE, "buffer %d is read-only / has not been created yet", buf);
                *p = q;
                *count = c->rows[jj] * c->K;
                return PMX_OK;
            }
            FAIL(PMX_E_UNSUPPORTED, "buffer %d does not exist in an fp64 context", buf);
    }
    *count = c->rows[j] * c->K;
    return PMX_OK;
}
extern "C" int pmx_upload_f64(pmx_ctx* c, int buf, const double* host, int64_t count) {
    if (!c || !host) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    double* d; int64_t n;
    int rc = buf_lookup_f64(c, buf, true, &d, &n);
    if (rc != PMX_OK) return rc;
    if (count != n) FAIL(PMX_E_INVALID, "buffer %d holds %lld doubles, got %lld", buf, (long long)n, (long long)count);
    HIP_CHECK(hipMemcpyAsync(d, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}
extern "C" int pmx_download_f64(pmx_ctx* c, int buf, double* host, int64_t count) {
    if (!c || !host) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    double* d; int64_t n;
    int rc = buf_lookup_f64(c, buf, false, &d, &n);
    if (rc != PMX_OK) return rc;
    if (count != n) FAIL(PMX_E_INVALID, "buffer %d holds %lld doubles, got %lld", buf, (long long)n, (long long)count);
    HIP_CHECK(hipMemcpyAsync(host, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
// building blocks of the chains
// ------------------------------------------------------------------------------------------------
static int read_status(pmx_ctx* c) {
    HIP_CHECK(hipMemcpyAsync(c->hstatus, c->dstatus, sizeof(DevStatus), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

static int reset_status(pmx_ctx* c) {
    DevStatus s;
    memset(&s, 0, sizeof(s));
    for (int f = 0; f < 2; ++f)
        for (int k = 0; k < MAXK; ++k) s.eigvec[f][k] = 1.0;
    memcpy(c->hstatus, &s, sizeof(s));
    HIP_CHECK(hipMemcpyAsync(c->dstatus, c->hstatus, sizeof(DevStatus), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

// what to tell the caller when a chain stopped with HALT_ERROR on a path that cannot repeat the iteration (one iteration per call)
static const char* chain_error_text(pmx_ctx* c) {
    if (c->hstatus && c->hstatus->k1_fault == 4)
        return "mode f16x2: K max|A| max|S| is more than 2^16 max|Y| -- one fp16 scale cannot carry this residual (f16_range_fault); create the context with PMX_MODE_F32";
    return "device chain reported an error";
}
// clear `halt` (and adaprox's need_sub flags) before resuming a chain
static int clear_halt(pmx_ctx* c) {
    static const int zeros[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(&c->dstatus->halt, zeros, sizeof(int) * 2, hipMemcpyHostToDevice, c->stream));   // halt, reason
    HIP_CHECK(hipMemcpyAsync(&c->dstatus->need_sub[0], zeros, sizeof(int) * 2, hipMemcpyHostToDevice, c->stream));
    return PMX_OK;
}

// Leave the chained gA accumulation of the fp16 K1 for good: one slab per column region, summed by the update kernels.
static int chain_disable(pmx_ctx* c) {
    if (c->chainL == 0) return PMX_OK;
    c->chainL = 0;
    c->nSlabA = c->plan.nSlabA;
    float* big = nullptr;
    int rc = dallocT(c, &big, (size_t)c->nSlabA * c->Mk * c->Kk, false);
    if (rc != PMX_OK) return rc;
    c->slab[0] = big;
    return PMX_OK;
}
// Entry points that run ONE iteration (or a piece of one) per call -- user callables, the line search, row-sharded bsdmm -- have
// nothing to repeat an iteration into: no chained K1 there (its faults are repaired by repeating), and the fp16 kernels' range
// guard is looked at right behind every K1 launch instead (enqueue_grad), where nothing else has been enqueued yet.
static int one_iteration_per_call(pmx_ctx* c) {
    c->k1_sync_check = true;
    return chain_disable(c);
}
// DevStatus::k1_fault == 4 (f16_range_fault, k_grad_f16_v8.hip): the two-term fp16 K1 refused a launch because one power-of-two
// scale cannot carry this residual.  Leave the fp16 kernels for good: the exact-fp32 K1 of the SAME frame (k_grad_f32_pc at
// K1's K = 32 / 64, k_grad_f32<128> else; zero-padded Y / factor copies stay as they are), with its own grid, slabs, loss
// partials and chain words.  The buffers of the fp16 plan stay allocated until the context is destroyed.
static int k1_leave_f16(pmx_ctx* c) {
    c->mode = PMX_MODE_F32;
    select_k1(c, c->Mk, c->Nk, c->ncu);
    c->fix_on = false;
    if (c->k1_sync_check) {                 // one-iteration-per-call paths run without chains (one_iteration_per_call: a chain fault could not be repaired by repeating the iteration)
        c->chainL = 0;
        c->nSlabA = c->plan.nSlabA;
    }
    c->chainFlags = nullptr;
    c->chainSeq = 0;
    c->slab[0] = c->slab[1] = nullptr;
    c->lossPart = nullptr;
    int rc = PMX_OK;
    if (c->chainL > 0) rc = dallocT(c, &c->chainFlags, (size_t)(c->plan.gridX * c->plan.gridY / c->chainL) * c->plan.RP * 4);
    if (rc == PMX_OK) rc = dallocT(c, &c->slab[0], (size_t)c->nSlabA * c->Mk * c->Kk, false);
    if (rc == PMX_OK) rc = dallocT(c, &c->slab[1], (size_t)c->plan.nSlabS * c->Nk * c->Kk, false);
    if (rc == PMX_OK) rc = dallocT(c, &c->lossPart, (size_t)2 * c->plan.gridX * c->plan.gridY);
    c->rangeFaults += 1;
    return rc;
}
// After read_status: a chained K1 launch found that its hand-off does not hold here (a predecessor on another XCD, or
// workgroups that are not co-resident: DevStatus::k1_fault) and stopped the chain of kernels before anything was updated.
// Fall back to slabs and clear the halt; the caller re-enqueues from DevStatus::it_done.  *again = 1 if that happened.
// The same for the fused adaprox tail (DevStatus::tail_fault: its census barrier timed out before anything was written).
static int chain_fault_fallback(pmx_ctx* c, int* again) {
    *again = 0;
    if (!c->hstatus->k1_fault && !c->hstatus->tail_fault) return PMX_OK;
    if (c->hstatus->tail_fault == 2)         // a barrier inside the fused tail never completed: the iteration is half applied
        FAIL(PMX_E_HIP, "k_ada_tail: a grid barrier timed out after the census had passed (a workgroup was lost); the factors are not usable");
    int rc = PMX_OK;
    if (c->hstatus->k1_fault == 4) {
        rc = k1_leave_f16(c);
        if (rc != PMX_OK) return rc;
    } else if (c->hstatus->k1_fault) {
        rc = chain_disable(c);
        if (rc != PMX_OK) return rc;
        c->chainFaults += 1;
    }
    if (c->hstatus->tail_fault) {
        c->tail_fused = false;
        c->tailFaults += 1;
    }
    static const int zeros[4] = {0, 0, 0, 0};
    HIP_CHECK(hipMemcpyAsync(&c->dstatus->tail_fault, zeros, 3 * sizeof(int), hipMemcpyHostToDevice, c->stream));   // tail_fault, pad2, k1_fault
    rc = clear_halt(c);
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->tickets) {                        // the update launches that followed the fault were skipped: start the count over
        HIP_CHECK(hipMemsetAsync(c->tickets, 0, sizeof(unsigned), c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->ticketLaunches = 0;
    }
    c->hstatus->k1_fault = 0;
    c->hstatus->tail_fault = 0;
    c->hstatus->halt = 0;
    c->hstatus->reason = 0;
    c->absmax_by_finish = false;
    c->gram_by_update = false;
    *again = 1;
    return PMX_OK;
}

static GradArgs small_grad_args(pmx_ctx* c, const float* A, const float* St, int doA, int doS) {
    GradArgs g{};
    g.Y = c->Y; g.ldY = c->ldY;
    g.W = c->W; g.ldW = c->ldW;
    g.A = A; g.St = St;
    g.slabA = c->slab[0]; g.slabS = c->slab[1];
    g.lossPart = c->lossPart;
    g.status = c->dstatus;
    g.M = (int)c->Mk; g.N = (int)c->Nk; g.K = (int)c->Kk;
    g.RP = c->plan.RP;
    g.doA = doA; g.doS = doS;
    return g;
}
static bool eig_small_applies(const pmx_ctx* c) { return c->K <= 16 && c->M <= 8192 && c->N <= 8192; }
static EigArgs small_eig_args(pmx_ctx* c, const float* A, const float* St, bool wantStepA, bool wantStepS, double scale) {
    EigArgs e{};
    e.G = c->gramG; e.Gw = c->gramG; e.KP = c->KP; e.K = (int)c->K; e.status = c->dstatus;
    e.want[0] = wantStepS; e.want[1] = wantStepA;     // factor 0 (A) -> step of block 1 (S), factor 1 (St) -> step of block 0 (A)
    e.scale = scale;
    e.max_iter = 200;
    e.Q = c->eigQ;
    e.X[0] = A; e.X[1] = St;
    e.rows[0] = c->M; e.rows[1] = c->N;
    return e;
}
// small problems: K1 and both step rules of a pgm iteration in ONE launch (k_small_front)
static int enqueue_small_front(pmx_ctx* c, const float* A, const float* St, double scale) {
    const bool timed = c->timing && (c->timing_seq++ % (unsigned)c->timing_stride) == 0 && c->ev_used + 2 <= c->ev.size();
    const GradArgs g = small_grad_args(c, A, St, 1, 1);
    const EigArgs e = small_eig_args(c, A, St, true, true, scale);
    if (timed) HIP_CHECK(hipEventRecord(c->ev[c->ev_used], c->stream));
    HIP_CHECK(launch_small_front(c->plan, g, e, c->stream));
    c->nloss = c->plan.gridX * c->plan.gridY;
    if (timed) {
        HIP_CHECK(hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_used += 2;
    }
    return PMX_OK;
}

// [r6] Consumer-wave priority per tuned K1 (same-box A/B, profiles/r06_b_setprio_ab.txt): K = 128 and K = 32 gain 1-2.5 % at level 1 (the consumers
// are the pole of the slot and the younger half of the workgroup), K = 64 shows nothing outside the noise and keeps 0.
static int k1_prio_for(const pmx_ctx* c, int kk) {
    if (c->k1_prio != -999) return c->k1_prio;
    return kk == 64 ? 0 : 1;
}

// absmax_fresh: the factor maxima in c->absmax were written by the kernel that produced A and St (k_ada_finish)
static int enqueue_gfix(pmx_ctx* c, const float* A, const float* St, int doA, int doS, hipStream_t stream) {
    GfixArgs f{};
    f.X[0] = A; f.X[1] = St;
    f.rows[0] = c->M; f.rows[1] = c->N;
    f.K = (int)c->Kk;
    f.absmax = c->absmax;
    f.part = c->fixPart; f.Q = c->fixQ;
    f.out[0] = c->fixSlab[0]; f.out[1] = c->fixSlab[1];
    f.ld = (int)c->Kk;
    f.want[0] = (doA & 1) != 0; f.want[1] = doS != 0;
    f.status = c->dstatus;
    f.prof = c->fixProf;
    HIP_CHECK(launch_gfix(f, stream));
    return PMX_OK;
}

static int enqueue_grad_once(pmx_ctx* c, const float* A, const float* St, int doA, int doS, bool absmax_fresh) {
    if (c->host_grad) return PMX_OK;       // a user `grad` callable: its result is already in G (see slab_ref)
    if (c->Kk != c->K) {                   // K1 runs the next tuned K: its operands are zero-padded copies of the factors
        PadArgs pa{};
        pa.src[0] = A; pa.src[1] = St;
        pa.dst[0] = c->Xk[0]; pa.dst[1] = c->Xk[1];
        pa.rows[0] = c->M; pa.rows[1] = c->N;
        pa.K = (int)c->K; pa.Kk = (int)c->Kk;
        pa.status = c->dstatus;
        launch_pad_factors(pa, c->stream);
        A = c->Xk[0]; St = c->Xk[1];
    }
    const int64_t Kq = c->Kk;              // K as K1 sees it (rows of A / St are Kq floats apart)
    bool fix_pending = false;              // an <HH> kernel was launched: its correction slab is owed (behind the timing events: they bracket K1 alone)
    const bool timed = c->timing && (c->timing_seq++ % (unsigned)c->timing_stride) == 0 && c->ev_used + 2 <= c->ev.size();
    if (c->k128) {
        AbsmaxArgs am{};
        am.X[0] = A; am.X[1] = St;
        am.count[0] = c->M * Kq; am.count[1] = c->N * Kq;
        am.out = c->absmax;
        am.status = c->dstatus;
        if (!absmax_fresh) launch_absmax(am, c->stream);
        SplitAArgs sp{};
        sp.X = A; sp.count = c->M * Kq; /* (a frame's extra rows stay zero) */ sp.absmax = c->absmax; sp.H = c->A16[0]; sp.L = c->A16[1]; sp.status = c->dstatus;
        launch_split_a_f16(sp, c->stream);
        GradK128Args g{};
        g.Y = c->Y; g.ldY = c->ldY;
        g.Ah = c->A16[0]; g.Al = c->A16[1]; g.St = St;
        g.slabA = c->slab[0]; g.slabS = c->slab[1];
        g.lossPart = c->lossPart;
        g.status = c->dstatus;
        g.M = (int)c->Mk; g.N = (int)c->Nk;
        g.RP = c->plan.RP;
        g.doA = doA; g.doS = doS;
        g.gridX = c->plan.gridX; g.gridY = c->plan.gridY;
        g.absmax = c->absmax; g.ymax = c->ymax;
        g.W = c->W; g.ldW = c->ldW; g.wmax = c->wmax;
        g.wstatus = c->dstatus; g.rangeRatio = c->rangeRatio;
        if (c->chainL > 0 && (doA & 1)) {    // k_grad_f16_k128<.., CHAIN>
            if (c->chainSeq >= (1u << 21)) {   // arrival words would run out of bits: start over
                HIP_CHECK(hipMemsetAsync(c->chainFlags, 0, (size_t)(c->plan.gridX * c->plan.gridY / c->chainL) * c->plan.RP * 4 * sizeof(unsigned), c->stream));
                c->chainSeq = 0;
            }
            g.chainL = c->chainL; g.chainStride = grad_k128_chain_stride(c->plan, c->chainL);
            g.chainFlags = c->chainFlags; g.chainBase = (++c->chainSeq) * 64u; g.wstatus = c->dstatus;
            g.chainInject = c->hook_inject_k1 > 0 && (int)c->chainSeq == c->hook_inject_k1;   // tests (read once, at pmx_ctx_create)
        }
        g.hh = c->f16_r3 == 2 && c->W == nullptr && c->fixPart != nullptr && ((doA & 1) || doS);     // (the launcher's own test: gradient passes only)
        g.consPrio = k1_prio_for(c, 128);
        fix_pending = g.hh != 0;
        if (timed) HIP_CHECK(hipEventRecord(c->ev[c->ev_used], c->stream));
        HIP_CHECK(grad_launch_k128(g, c->stream));
        c->nloss = c->plan.gridX * c->plan.gridY;
    } else if (c->k32f16) {
        AbsmaxArgs am{};
        am.X[0] = A; am.X[1] = St;
        am.count[0] = c->M * Kq; am.count[1] = c->N * Kq;
        am.out = c->absmax;
        am.status = c->dstatus;
        if (!absmax_fresh) launch_absmax(am, c->stream);
        GradV4Args g{};
        g.Y = c->Y; g.ldY = c->ldY;
        g.A = A; g.St = St;
        g.slabA = c->slab[0]; g.slabS = c->slab[1];
        g.lossPart = c->lossPart;
        g.status = c->dstatus;
        g.M = (int)c->Mk; g.N = (int)c->Nk;
        g.RP = c->plan.RP;
        g.doA = doA; g.doS = doS;
        g.gridX = c->plan.gridX; g.gridY = c->plan.gridY;
        g.absmax = c->absmax; g.ymax = c->ymax; g.wmax = 1.f;
        g.wstatus = c->dstatus; g.rangeRatio = c->rangeRatio;
        g.r3 = c->f16_r3;
        g.consPrio = k1_prio_for(c, 32);
        g.fold = c->k1_fold; c->k1_fold = K1GramFold{};
        if (timed) HIP_CHECK(hipEventRecord(c->ev[c->ev_used], c->stream));
        HIP_CHECK(grad_launch_f16_k32(g, c->stream));
        c->nloss = c->plan.gridX * c->plan.gridY;
    } else if (c->use_bf16) {
        PresplitArgs ps{};
        ps.X[0] = A; ps.X[1] = St;
        for (int j = 0; j < 2; ++j) { ps.Xp[j] = c->Bp[j]; ps.Xt[j] = c->Bt[j]; ps.rows[j] = c->rowsK[j]; ps.rowsPad[j] = c->rowsPad[j]; }
        ps.K = (int)Kq; ps.KP = c->KP;
        ps.status = c->dstatus;
        if (!grad_bf16_reads_fp32(c->plan, c->Mk, c->Nk, Kq)) launch_presplit(ps, c->stream);
        GradBfArgs g{};
        g.Y = c->Y; g.ldY = c->ldY;
        g.Ap = c->Bp[0]; g.At = c->Bt[0]; g.Sp = c->Bp[1]; g.Stt = c->Bt[1];
        g.MPad = c->rowsPad[0]; g.NPad = c->rowsPad[1];
        g.slabA = c->slab[0]; g.slabS = c->slab[1];
        g.lossPart = c->lossPart;
        g.status = c->dstatus;
        g.M = (int)c->Mk; g.N = (int)c->Nk; g.K = (int)Kq;
        g.RP = c->plan.RP;
        g.doA = doA; g.doS = doS;
        g.prof = c->k1prof;
        g.W = c->W; g.ldW = c->ldW;
        if (c->use_f16) {   // operand scales of the two-term fp16 kernel: maxima of THESE factor arrays
            AbsmaxArgs am{};
            am.X[0] = A; am.X[1] = St;
            am.count[0] = c->M * Kq; am.count[1] = c->N * Kq;
            am.out = c->absmax;
            am.status = c->dstatus;
            if (!absmax_fresh) launch_absmax(am, c->stream);
            g.absmax = c->absmax; g.ymax = c->ymax; g.wmax = c->wmax;
            g.wstatus = c->dstatus; g.rangeRatio = c->rangeRatio; g.r3 = c->f16_r3; g.consPrio = k1_prio_for(c, 64);
        }
        if (c->chainL > 0) {                 // k_grad_f16_v8<.., CHAIN> / k_grad_bf16_v7<.., CHAIN>
            if (c->chainSeq >= (1u << 21)) {   // arrival words would run out of bits: start over
                HIP_CHECK(hipMemsetAsync(c->chainFlags, 0, (size_t)(c->plan.gridX * c->plan.gridY / c->chainL) * c->plan.RP * 4 * sizeof(unsigned), c->stream));
                c->chainSeq = 0;
            }
            g.chainL = c->chainL; g.chainFlags = c->chainFlags; g.chainBase = (++c->chainSeq) * 64u; g.wstatus = c->dstatus;
            g.chainInject = c->hook_inject_k1 > 0 && (int)c->chainSeq == c->hook_inject_k1;   // tests (read once, at pmx_ctx_create)
        }
        if (timed) HIP_CHECK(hipEventRecord(c->ev[c->ev_used], c->stream));
        bool took_f16 = false;
        HIP_CHECK(grad_launch_bf16(c->plan, g, A, St, c->stream, &c->nloss, &took_f16));
        c->f16_fell_back = c->use_f16 && !took_f16;      // (pmx_k1_info reports the kernel that ran)
        if ((doA & 1) || doS) fix_pending = c->use_f16 && took_f16 && c->f16_r3 == 2 && c->W == nullptr && c->fixPart != nullptr;   // (grad_launch_f16_v8's own test)
    } else {
        GradArgs g = small_grad_args(c, A, St, doA, doS);
        if (c->f32pc && !c->use_small) { g.fold = c->k1_fold; c->k1_fold = K1GramFold{}; }     // (k_grad_f32_pc carries the fold; pgm_enqueue_iteration arms it for that kernel only)
        if (timed) HIP_CHECK(hipEventRecord(c->ev[c->ev_used], c->stream));
        if (c->f32pc && c->chainL > 0) {
            if (c->chainSeq >= (1u << 21)) {   // arrival words would run out of bits: start over
                HIP_CHECK(hipMemsetAsync(c->chainFlags, 0, (size_t)(c->plan.gridX * c->plan.gridY / c->chainL) * c->plan.RP * 4 * sizeof(unsigned), c->stream));
                c->chainSeq = 0;
            }
            GradArgs gc = g;
            gc.chainL = c->chainL; gc.chainFlags = c->chainFlags; gc.chainBase = (++c->chainSeq) * 64u; gc.wstatus = c->dstatus;
            gc.chainInject = c->hook_inject_k1 > 0 && (int)c->chainSeq == c->hook_inject_k1;   // tests (read once, at pmx_ctx_create)
            HIP_CHECK(grad_launch_f32pc(c->plan, gc, c->stream));
        } else
        HIP_CHECK(c->use_small ? grad_launch_small(c->plan, g, c->stream) : (c->f32pc ? grad_launch_f32pc(c->plan, g, c->stream) : grad_launch_f32(c->plan, g, c->stream)));
        c->nloss = c->plan.gridX * c->plan.gridY;
    }
    if (timed) {
        HIP_CHECK(hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_used += 2;
    }
    if ((doA & 1) || doS) c->fix_on = fix_pending;
    // what the high x high residual left out, as one more slab per block (k_gfix.hip).  In the launch stream: the three launches depend on the
    // factors only, but nothing fits BESIDE K1 (its waves hold all 512 registers of every SIMD) -- measured on a stream of their own they sat
    // behind K1's workgroups and the pass got 2 % slower (profiles/r05_a_gfix_side_stream.txt)
    if (fix_pending) { const int rcf = enqueue_gfix(c, A, St, doA, doS, c->stream); if (rcf != PMX_OK) return rcf; }
    return PMX_OK;
}

static int enqueue_grad(pmx_ctx* c, const float* A, const float* St, int doA, int doS, bool absmax_fresh = false) {
    int rc = enqueue_grad_once(c, A, St, doA, doS, absmax_fresh);
    if (rc != PMX_OK || !c->k1_sync_check || !c->f16_scales || c->host_grad || !(doA | doS)) return rc;
    // a path with nothing to repeat an iteration into: the fp16 kernel's range guard (f16_range_fault) is looked at NOW, with nothing
    // behind the launch yet -- refused: the context leaves the fp16 kernels and the same gradient pass runs in exact fp32
    rc = read_status(c);
    if (rc != PMX_OK) return rc;
    if (c->hstatus->k1_fault != 4) return PMX_OK;
    int again = 0;
    rc = chain_fault_fallback(c, &again);
    if (rc != PMX_OK) return rc;
    return enqueue_grad_once(c, A, St, doA, doS, false);
}

static SlabRef slab_ref(pmx_ctx* c, int j) {
    SlabRef s;
    s.stride = c->rowsK[j] * c->Kk;
    s.ld = (int)c->Kk;
    if (c->host_grad) {      // the caller's gradient: ONE "slab", the G buffer itself (the update kernels fold it onto itself)
        s.base = c->G[j];
        s.n = 1;
        s.ld = (int)c->K;
        return s;
    }
    s.base = c->slab[j];
    s.n = j == 0 ? c->nSlabA : c->nSlabS;
    if (c->fix_on) s.extra = c->fixSlab[j];
    return s;
}

// Gram matrices + largest eigenvalues -> DevStatus::step.  wantA: step of block 0 (needs factor 1 = St)
static int bsdmm_flush_decide(pmx_ctx* c) {
    if (!c->bsd_decide_pending) return PMX_OK;
    launch_bsdmm_decide(c->bsd_decide, c->stream);
    HIP_CHECK(hipGetLastError());
    c->bsd_decide_pending = false;
    return PMX_OK;
}
static int enqueue_steps(pmx_ctx* c, const float* A, const float* St, bool wantStepA, bool wantStepS, double scale, bool have_partials = false, bool with_decide = false) {
    if (eig_small_applies(c)) {
        int frc = bsdmm_flush_decide(c);
        if (frc != PMX_OK) return frc;       // small factors: Gram + reduce + lmax in ONE launch (k_eig_small forms G itself)
        const EigArgs e = small_eig_args(c, A, St, wantStepA, wantStepS, scale);
        HIP_CHECK(launch_eig(e, c->stream));
        return PMX_OK;
    }
    GramArgs g{};
    g.X[0] = A; g.X[1] = St;
    g.rows[0] = c->M; g.rows[1] = c->N;
    g.K = (int)c->K;
    g.part = c->gramPart;
    g.status = c->dstatus;
    g.want[0] = wantStepS;   // factor 0 (A)  -> step of block 1 (S)
    g.want[1] = wantStepA;   // factor 1 (St) -> step of block 0 (A)
    if (!have_partials) launch_gram(g, c->KP, c->stream);     // (have_partials: k_pgm_update left them, PgmArgs::gramPart)
    GramReduceArgs r{};
    r.part = c->gramPart; r.G = c->gramG; r.KP = c->KP; r.status = c->dstatus;
    r.want[0] = g.want[0]; r.want[1] = g.want[1];
    r.nparts[0] = gram_nparts(c->M); r.nparts[1] = gram_nparts(c->N);
    if (with_decide) {                       // pgm: the previous iteration's stopping test as one more workgroup of this launch
        r.dec_partials = c->partials; r.dec_status = c->dstatus;
        r.dec_e_rel[0] = c->pgm.e_rel[0]; r.dec_e_rel[1] = c->pgm.e_rel[1];
    }
    if (c->bsd_decide_pending) {             // bsdmm: the Boyd test of the block updated just before, likewise
        r.dec_bsdmm = c->bsd_decide;
        c->bsd_decide_pending = false;
    }
    launch_gram_reduce(r, c->stream);
    EigArgs e{};
    e.G = c->gramG; e.KP = c->KP; e.K = (int)c->K; e.status = c->dstatus;
    e.want[0] = g.want[0]; e.want[1] = g.want[1];
    e.scale = scale;
    e.max_iter = 200;
    e.Q = c->eigQ;
    HIP_CHECK(launch_eig(e, c->stream));
    return PMX_OK;
}

static int enqueue_gram_only(pmx_ctx* c, const float* A, const float* St, bool wantA_factor, bool wantS_factor) {
    GramArgs g{};
    g.X[0] = A; g.X[1] = St;
    g.rows[0] = c->M; g.rows[1] = c->N;
    g.K = (int)c->K;
    g.part = c->gramPart;
    g.status = c->dstatus;
    g.want[0] = wantA_factor; g.want[1] = wantS_factor;
    launch_gram(g, c->KP, c->stream);
    GramReduceArgs r{};
    r.part = c->gramPart; r.G = c->gramG; r.KP = c->KP; r.status = c->dstatus;
    r.want[0] = g.want[0]; r.want[1] = g.want[1];
    r.nparts[0] = gram_nparts(c->M); r.nparts[1] = gram_nparts(c->N);
    launch_gram_reduce(r, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
static int enqueue_eig_only(pmx_ctx* c, bool wantA_factor, bool wantS_factor, double scale) {
    EigArgs e{};
    e.G = c->gramG; e.KP = c->KP; e.K = (int)c->K; e.status = c->dstatus;
    e.want[0] = wantA_factor; e.want[1] = wantS_factor;
    e.scale = scale;
    e.max_iter = 200;
    e.Q = c->eigQ;
    HIP_CHECK(launch_eig(e, c->stream));
    return PMX_OK;
}
static int shard_gram_in(pmx_ctx* c) {
    GramInArgs g{};
    g.comm_gram = c->ssplit ? c->comm_out + c->sncol * c->K : c->comm + c->N * c->K;     // (S-split: this rank's reduced chunk)
    g.G = c->gramG;
    g.n = c->KP * c->KP;
    g.status = c->dstatus;
    g.peer_halt = c->algo == ALG_BSDMM ? c->comm + c->N * c->K + (int64_t)c->KP * c->KP + MAXK + SHARD_HALT_SLOT : nullptr;
    g.wstatus = c->dstatus;
    launch_shard_gram_in(g, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// f64_ok: the entry point has an fp64 implementation (include/pmx.h: PMX_MODE_F64 lists them); every other one refuses an
// fp64 context instead of touching float arrays it does not have
static int require_ready(pmx_ctx* c, bool f64_ok = false) {
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (c->f64 && !f64_ok) FAIL(PMX_E_UNSUPPORTED, "this entry point has no fp64 implementation (PMX_MODE_F64 covers the fused pgm / FISTA, adaprox and bsdmm loops on small problems)");
    if (!c->haveY) FAIL(PMX_E_STATE, "Y has not been set");
    HIP_CHECK(hipSetDevice(c->device));
    return PMX_OK;
}

static void fill_result(pmx_ctx* c, pmx_result* r, int it_before) {
    if (!r) return;
    const DevStatus* s = c->hstatus;
    r->iterations = s->it_done - it_before;
    r->total_iterations = s->it_done;
    r->stopped = s->stopped;
    r->converged[0] = s->conv[0];
    r->converged[1] = s->conv[1];
    r->steps[0] = s->step[0];
    r->steps[1] = s->step[1];
    r->sub_iterations[0] = s->sub_total[0];
    r->sub_iterations[1] = s->sub_total[1];
}

// ------------------------------------------------------------------------------------------------
// PMX_MODE_F64: K1 (+ the step rule) and the pgm iteration of an fp64 context
// ------------------------------------------------------------------------------------------------
// [r6] k_big_f64.hip: one MFMA pass per gradient wanted (gSt first: it carries the loss), then the step rule in three launches
static int enqueue_front64_big(pmx_ctx* c, const double* A, const double* St, int doA, int doS, bool tiles, bool steps, double scale) {
    if (tiles) {
        const bool loss_only = !doA && !doS;
        const double* X[2] = {A, St};
        if (c->Xk64[0] || c->Xk64[1]) {              // the padded copies K1 reads (rows to 64, K to KP)
            Pad64Args pa{};
            for (int j = 0; j < 2; ++j) { pa.X[j] = X[j]; pa.P[j] = c->Xk64[j]; pa.rows[j] = c->Xk64[j] ? c->rows[j] : 0; }
            pa.K = (int)c->K; pa.KP = c->KP; pa.status = c->dstatus;
            launch_pad64(pa, c->stream);
            for (int j = 0; j < 2; ++j) if (c->Xk64[j]) X[j] = c->Xk64[j];
        }
        for (int j = 1; j >= 0; --j) {
            const bool want = j ? (doS || loss_only) : (doA != 0);
            if (!want) continue;
            Pass64Args p{};
            p.Y = c->Yd; p.ldY = c->ldY64;
            p.Wt = c->Wd_on ? c->Wd : nullptr;
            p.F = X[j]; p.W = X[1 - j];
            p.rowsF = (int)c->rows[j]; p.rowsW = (int)c->rows[1 - j];
            p.K = (int)c->K;
            p.slab = c->slabd[j];
            p.status = c->dstatus;
            p.nsplit = c->nsplit64[j]; p.bps = c->bps64[j];
            p.store = !loss_only;
            const bool with_loss = j == 1 || !doS;           // (the loss rides in gSt's pass, or in gA's when that is the only one)
            p.lossPart = with_loss ? c->lossPart : nullptr;
            HIP_CHECK(launch_grad64_pass(p, c->KP, j == 0, c->stream));
            if (with_loss) c->nloss = j ? c->t64y : c->t64x;
        }
    }
    if (steps) {
        Gram64Args g{};
        g.X[0] = A; g.X[1] = St;
        g.rows[0] = c->M; g.rows[1] = c->N;
        g.K = (int)c->K;
        g.part = c->gramPart64; g.G = c->gramG;
        g.status = c->dstatus;
        g.want[0] = 1; g.want[1] = 1;
        launch_gram64(g, c->KP, c->stream);
        EigArgs e{};
        e.G = c->gramG; e.Gw = c->gramG; e.KP = c->KP; e.K = (int)c->K; e.status = c->dstatus;
        e.want[0] = 1; e.want[1] = 1;                // factor 0 (A) -> step of block 1 (S), factor 1 (St) -> step of block 0 (A)
        e.scale = scale;
        e.max_iter = 200;
        e.Q = c->eigQ;
        e.force_exact = 2;                           // lambda_max to fp64 round-off by power steps on the fp64 matrix (k_gram.hip: EigArgs::force_exact)
        HIP_CHECK(launch_eig(e, c->stream));
    }
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
static int enqueue_front64(pmx_ctx* c, const double* A, const double* St, int doA, int doS, bool tiles, bool steps, double scale) {
    if (c->f64big) return enqueue_front64_big(c, A, St, doA, doS, tiles, steps, scale);
    Grad64Args g{};
    g.Y = c->Yd; g.ldY = c->N;
    g.A = A; g.St = St;
    g.slabA = c->slabd[0]; g.slabS = c->slabd[1];
    g.lossPart = c->lossPart;
    g.status = c->dstatus;
    g.M = (int)c->M; g.N = (int)c->N; g.K = (int)c->K;
    g.doA = doA; g.doS = doS;
    EigArgs e{};
    e.G = c->gramG; e.Gw = c->gramG; e.KP = c->KP; e.K = (int)c->K; e.status = c->dstatus;
    e.want[0] = steps; e.want[1] = steps;        // factor 0 (A) -> step of block 1 (S), factor 1 (St) -> step of block 0 (A)
    e.scale = scale;
    e.max_iter = 200;
    e.Q = c->eigQ;
    e.X[0] = nullptr; e.X[1] = nullptr;          // (the fp64 rows travel as their own arguments)
    e.force_exact = 1;                           // lambda_max to fp64 round-off (k_gram.hip: EigArgs::force_exact)
    e.rows[0] = c->M; e.rows[1] = c->N;
    HIP_CHECK(launch_front64(g, e, A, St, tiles ? c->t64x : 0, tiles ? c->t64y : 0, steps, c->stream));
    if (tiles) c->nloss = c->t64x * c->t64y;
    return PMX_OK;
}
static int pgm64_enqueue_iteration(pmx_ctx* c) {
    const pmx_pgm_params& p = c->pgm;
    const double* A = p.accelerated ? c->Xed[0] : c->Xd[0];
    const double* St = p.accelerated ? c->Xed[1] : c->Xd[1];
    int rc = enqueue_front64(c, A, St, 1, 1, true, !p.use_fixed_steps, (double)p.step_scale);   // algorithms.py:105-106
    if (rc != PMX_OK) return rc;
    Pgm64Args u{};
    for (int j = 0; j < 2; ++j) {
        u.X[j] = c->Xd[j];
        u.Xe[j] = p.accelerated ? c->Xed[j] : c->Xd[j];
        u.G[j] = c->Gd[j];
        u.slab[j] = c->slabd[j];
        u.nslab[j] = j == 0 ? c->nSlabA : c->nSlabS;
        u.rows[j] = c->rows[j];
        u.prox[j] = to_dev(p.prox[j]);
    }
    u.K = (int)c->K;
    u.status = c->dstatus;
    u.partials = c->partials;
    u.accelerated = p.accelerated;
    {   // omega the NEXT iteration reads (utils.py:198-206), in fp64
        double om = 0.0;
        if (p.accelerated) {
            const double t = c->nest_t, t1 = 0.5 * (1.0 + sqrt(4.0 * t * t + 1.0));
            om = (t - 1.0) / t1;
            c->nest_t = t1;
        }
        u.omega_next = om;
    }
    const int64_t rmax = c->rows[0] > c->rows[1] ? c->rows[0] : c->rows[1];
    const int nbx = (int)((rmax + EW_THREADS / 32 - 1) / (EW_THREADS / 32));      // <= 256: M, N <= 8192
    if (c->f64big) launch_pgm64b_update(u, c->stream);
    else launch_pgm64_update(u, nbx, c->stream);                                   // algorithms.py:107-108
    DecideArgs d{};
    d.status = c->dstatus; d.partials = c->partials;
    d.e_rel[0] = p.e_rel[0]; d.e_rel[1] = p.e_rel[1];
    d.check = 1;
    launch_pgm_decide(d, c->stream);                                               // algorithms.py:130-135
    HIP_CHECK(hipGetLastError());
    c->it += 1;
    return PMX_OK;
}

// [r6] one pgm iteration with the Beck-Teboulle line search (algorithms.py:93-135) in fp64: host-driven -- every trial ends in an evaluation
// of the likelihood whose value decides the next step, so the chain cannot run ahead anyway
static int loss64_now(pmx_ctx* c, const double* A, const double* St, double* out) {
    int rc = enqueue_front64(c, A, St, 0, 0, true, false, 1.0);
    if (rc != PMX_OK) return rc;
    std::vector<double> h(c->nloss);
    HIP_CHECK(hipMemcpyAsync(h.data(), c->lossPart, c->nloss * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    double s = 0.0;
    for (double v : h) s += v;
    *out = 0.5 * s;
    return PMX_OK;
}
static int pgm64_bt_iteration(pmx_ctx* c) {
    const pmx_pgm_params& p = c->pgm;
    int rc = PMX_OK;
    if (c->it == 0) {                                            // f_prev = f(X_) (algorithms.py:113-114)
        rc = loss64_now(c, c->Xd[0], c->Xd[1], &c->bt_fprev);
        if (rc != PMX_OK) return rc;
    }
    for (int j = 0; j < 2; ++j)                                  // X_ = copy of X (:102)
        HIP_CHECK(hipMemcpyAsync(c->Xprevd[j], c->Xd[j], c->rows[j] * c->K * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    rc = enqueue_front64(c, c->Xed[0], c->Xed[1], 1, 1, true, !p.use_fixed_steps, (double)p.step_scale);      // G, S at _X (:105-106)
    if (rc != PMX_OK) return rc;
    Fold64Args f{};
    for (int j = 0; j < 2; ++j) { f.slab[j] = c->slabd[j]; f.nslab[j] = j == 0 ? c->nSlabA : c->nSlabS; f.G[j] = c->Gd[j]; f.count[j] = c->rows[j] * c->K; }
    launch_fold64b(f, c->stream);
    Bt64Args u{};
    for (int j = 0; j < 2; ++j) {
        u.X[j] = c->Xd[j]; u.E[j] = c->Xed[j]; u.Xp[j] = c->Xprevd[j]; u.G[j] = c->Gd[j];
        u.rows[j] = c->rows[j];
        u.prox[j] = to_dev(p.prox[j]);
        u.do_block[j] = 1;
    }
    u.K = (int)c->K;
    u.status = c->dstatus;
    u.partials = c->partials;
    std::vector<double> hp((size_t)SL_COUNT * 2 * EW_BLOCKS);
    double sums[2][5] = {};                                      // per block: (X - X_).G, (X - X_)^2, X^2, max |G|, max |X_|
    double f_now = 0.0;
    for (int trial = 0;; ++trial) {
        u.T[0] = c->btT[0]; u.T[1] = c->btT[1];
        launch_bt64_update(u, c->stream);                        // :108 / :125
        HIP_CHECK(hipGetLastError());
        rc = loss64_now(c, c->Xd[0], c->Xd[1], &f_now);          // f(*X) (:112, :126)
        if (rc != PMX_OK) return rc;
        HIP_CHECK(hipMemcpyAsync(hp.data(), c->partials, hp.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        auto slot = [&](int sl, int j) { return hp.data() + ((size_t)sl * 2 + j) * EW_BLOCKS; };
        for (int j = 0; j < 2; ++j) {
            if (!u.do_block[j]) continue;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, m0 = 0.0, m1 = 0.0;
            for (int b = 0; b < EW_BLOCKS; ++b) {                // fixed order
                s0 += slot(SL_BT0, j)[b]; s1 += slot(SL_DIFF2, j)[b]; s2 += slot(SL_NORM2, j)[b];
                const double g = slot(SL_BT0 + 1, j)[b], x = slot(SL_BT0 + 2, j)[b];
                m0 = (g != g || m0 != m0) ? NAN : std::max(m0, g);
                m1 = (x != x || m1 != m1) ? NAN : std::max(m1, x);
            }
            sums[j][0] = s0; sums[j][1] = s1; sums[j][2] = s2; sums[j][3] = m0; sums[j][4] = m1;
        }
        double q = 0.0;
        for (int j = 0; j < 2; ++j) q += sums[j][0] + 0.5 / (c->btT[j] * c->hstatus->step[j]) * sums[j][1];      // Beck & Teboulle, eq. 3.2 (:117-118)
        if (!(f_now > c->bt_fprev + q)) break;
        if (trial > 1100) FAIL(PMX_E_STATE, "pgm (fp64): the line search does not terminate");       // (T underflows to 0 after 1075 halvings)
        const double r0 = c->hstatus->step[0] * sums[0][3] / sums[0][4], r1 = c->hstatus->step[1] * sums[1][3] / sums[1][4];
        const int jm = (r1 > r0) ? 1 : 0;                        // np.argmax: the first of equals; a NaN in front wins (:121)
        const int jmax = (r0 != r0) ? 0 : ((r1 != r1) ? 1 : jm);
        c->btT[jmax] /= 2;                                       // :122
        u.do_block[0] = jmax == 0; u.do_block[1] = jmax == 1;
    }
    c->bt_fprev = f_now;                                         // :127
    // the stopping test (:130-135) from the sums of the accepted trial: k_pgm_decide folds the same slots
    DecideArgs d{};
    d.status = c->dstatus; d.partials = c->partials;
    d.e_rel[0] = p.e_rel[0]; d.e_rel[1] = p.e_rel[1];
    d.check = 1;
    launch_pgm_decide(d, c->stream);
    BtFin64Args fin{};
    double om = 0.0;
    if (p.accelerated) {                                         // omega the NEXT iteration reads (utils.py:198-206)
        const double t = c->nest_t, t1 = 0.5 * (1.0 + sqrt(4.0 * t * t + 1.0));
        om = (t - 1.0) / t1;
        c->nest_t = t1;
    }
    for (int j = 0; j < 2; ++j) { fin.X[j] = c->Xd[j]; fin.Xp[j] = c->Xprevd[j]; fin.E[j] = c->Xed[j]; fin.count[j] = c->rows[j] * c->K; }
    fin.omega = om;
    fin.status = c->dstatus;
    launch_bt64_finish(fin, c->stream);
    HIP_CHECK(hipGetLastError());
    c->it += 1;
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
// single operations
// ------------------------------------------------------------------------------------------------
extern "C" int pmx_grad(pmx_ctx* c) {
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (c->f64) {
        HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
        rc = enqueue_front64(c, c->Xd[0], c->Xd[1], 1, 1, true, false, 1.0);
        if (rc != PMX_OK) return rc;
        Fold64Args f{};
        for (int j = 0; j < 2; ++j) { f.slab[j] = c->slabd[j]; f.nslab[j] = j == 0 ? c->nSlabA : c->nSlabS; f.G[j] = c->Gd[j]; f.count[j] = c->rows[j] * c->K; }
        if (c->f64big) launch_fold64b(f, c->stream);
        else launch_fold64(f, c->stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return PMX_OK;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
        rc = enqueue_grad(c, c->X[0], c->X[1], 1, 1);
        if (rc != PMX_OK) return rc;
        FoldArgs f{};
        for (int j = 0; j < 2; ++j) { f.slab[j] = slab_ref(c, j); f.G[j] = c->G[j]; f.rows[j] = c->rows[j]; }
        f.K = (int)c->K;
        f.status = c->dstatus;
        launch_fold(f, 2, c->stream);
        HIP_CHECK(hipGetLastError());
        if (c->chainL == 0 && !c->f16_scales) { HIP_CHECK(hipStreamSynchronize(c->stream)); break; }
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        int again = 0;
        rc = chain_fault_fallback(c, &again);
        if (rc != PMX_OK) return rc;
        if (!again) break;
    }
    return PMX_OK;
}

extern "C" int pmx_time_grad(pmx_ctx* c, int do_A, int do_S, int reps, double* avg_ms) {
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (!avg_ms || reps < 1) FAIL(PMX_E_INVALID, "bad argument");
    if (getenv("PMX_K1_PROF") && !c->k1prof) {
        rc = dallocT(c, &c->k1prof, 16);
        if (rc != PMX_OK) return rc;
    }
    if (c->k1prof) HIP_CHECK(hipMemsetAsync(c->k1prof, 0, 16 * sizeof(unsigned long long), c->stream));
    HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
    const bool was = c->timing;
    c->timing = false;
    rc = enqueue_grad(c, c->X[0], c->X[1], do_A, do_S);   // warm-up
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps && rc == PMX_OK; ++i) rc = enqueue_grad(c, c->X[0], c->X[1], do_A, do_S);
    HIP_CHECK(hipEventRecord(e1, c->stream));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->timing = was;
    *avg_ms = ms / reps;
    if (c->fixProf && c->fix_on) {
        long long h[16];
        HIP_CHECK(hipMemcpy(h, c->fixProf, sizeof(h), hipMemcpyDeviceToHost));
        auto us = [&](int i, int j) { return (double)(h[j] - h[i]) * 0.01; };
        fprintf(stderr, "[gfixprof] workgroup 0, us: gram: start->scale %.2f, ->MFMAs done %.2f, ->stored %.2f | gram end -> reduce start %.2f | reduce %.2f | -> apply start %.2f | apply: ->scale %.2f, ->staged %.2f, ->task done %.2f, ->end %.2f | gram start -> apply end %.2f\n",
                us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(8, 9), us(9, 10), us(0, 10));
    }
    if (c->k1prof) {
        unsigned long long h[16];
        HIP_CHECK(hipMemcpy(h, c->k1prof, sizeof(h), hipMemcpyDeviceToHost));
        const double nwg = (double)c->plan.gridX * c->plan.gridY * (reps + 1);
        static const char* nm[10] = {"p0", "p1", "p2", "p3", "p4", "p5", "p6", "p7", "p8", "p9"};   // phase meaning: see the PH(i) marks of the kernel variant in use
        fprintf(stderr, "[k1prof] doA=%d doS=%d cycles per workgroup (wave 0):", do_A, do_S);
        for (int i = 0; i < 10; ++i) fprintf(stderr, " %s=%.0f", nm[i], (double)h[i] / nwg);
        fprintf(stderr, "\n");
    }
    return rc;
}

extern "C" int pmx_loglike(pmx_ctx* c, double* out) {
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (!out) FAIL(PMX_E_INVALID, "out is NULL");
    if (c->f64) {
        HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
        rc = enqueue_front64(c, c->Xd[0], c->Xd[1], 0, 0, true, false, 1.0);
        if (rc != PMX_OK) return rc;
        std::vector<double> h(c->nloss);
        HIP_CHECK(hipMemcpyAsync(h.data(), c->lossPart, c->nloss * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        double s = 0.0;
        for (double v : h) s += v;
        *out = 0.5 * s;
        return PMX_OK;
    }
    if (c->host_grad) FAIL(PMX_E_UNSUPPORTED, "pmx_loglike: this context runs on a caller-supplied gradient (pmx_set_host_grad) and has no Y to evaluate the likelihood on");
    HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
    rc = enqueue_grad(c, c->X[0], c->X[1], 0, 0);
    if (rc != PMX_OK) return rc;
    const int n = c->nloss;
    std::vector<double> h(n);
    HIP_CHECK(hipMemcpyAsync(h.data(), c->lossPart, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += h[i];
    *out = 0.5 * s;
    return PMX_OK;
}

extern "C" int pmx_step_pgm(pmx_ctx* c, double out[2]) {
    if (!c || !out) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    int rc = PMX_OK;
    // a context without a solver is a FUNCTION of its factors (nmf.step_pgm called by the caller's own code, on a context the host wrapper
    // keeps between calls): the power iteration starts where a fresh context's does, not from the previous call's eigenvector
    if (c->algo == ALG_NONE) { rc = reset_status(c); if (rc != PMX_OK) return rc; }
    HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
    rc = c->f64 ? enqueue_front64(c, c->Xd[0], c->Xd[1], 0, 0, false, true, 1.0) : enqueue_steps(c, c->X[0], c->X[1], true, true, 1.0);
    if (rc != PMX_OK) return rc;
    rc = read_status(c);
    if (rc != PMX_OK) return rc;
    out[0] = c->hstatus->step[0];
    out[1] = c->hstatus->step[1];
    return PMX_OK;
}

// S-split: the block-1 arrays the update kernels see are this rank's rows of S^T only
static float* s_view(pmx_ctx* c, int j, float* p) { return (p && j == 1 && c->ssplit) ? p + c->scol0 * c->K : p; }
static int64_t upd_rows(pmx_ctx* c, int j) { return (j == 1 && c->ssplit) ? c->sncol : c->rows[j]; }
static int64_t split_chunk(pmx_ctx* c) { return c->sncol * c->K + (int64_t)c->KP * c->KP + 2 * MAXK + 32; }

static AlphaArgs alpha_args(pmx_ctx* c) {
    AlphaArgs a{};
    a.status = c->dstatus;
    a.colpart = c->colpart;
    a.rows_global[0] = c->M_global;
    a.rows_global[1] = c->N;
    a.K = (int)c->K;
    a.use_fixed = 0;
    return a;
}

static int enqueue_alpha_from_factors(pmx_ctx* c, const AlphaArgs& al) {
    ColsumArgs cs{};
    cs.X[0] = c->X[0]; cs.X[1] = s_view(c, 1, c->X[1]);        // (S-split: this rank's columns; the partial sums meet in the collective)
    cs.rows[0] = c->M; cs.rows[1] = upd_rows(c, 1);
    cs.K = (int)c->K;
    cs.colpart = c->colpart;
    cs.status = c->dstatus;
    launch_colsum(cs, c->stream);
    launch_alpha_init(al, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// entry points without an fp64 implementation that do not pass through require_ready()
#define REJECT_F64(c) do { if ((c) && (c)->f64) FAIL(PMX_E_UNSUPPORTED, "%s has no fp64 implementation (PMX_MODE_F64 covers the fused pgm / FISTA, adaprox and bsdmm loops on small problems)", __func__); } while (0)

extern "C" int pmx_step_adaprox(pmx_ctx* c, float* out) {
    REJECT_F64(c);
    if (!c || !out) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, sizeof(int), c->stream));
    AlphaArgs al = alpha_args(c);
    if (c->algo == ALG_ADAPROX && c->ada.use_fixed_steps == 1) {   // inside a constant_step run the constants ARE the rule
        al.use_fixed = 1;
        al.fixed[0] = (float)c->ada.fixed_alpha[0];
        al.fixed[1] = (float)c->ada.fixed_alpha[1];
    }
    int rc = enqueue_alpha_from_factors(c, al);
    if (rc != PMX_OK) return rc;
    rc = read_status(c);
    if (rc != PMX_OK) return rc;
    memcpy(out, c->hstatus->alpha[0], sizeof(float) * c->K);
    memcpy(out + c->K, c->hstatus->alpha[1], sizeof(float) * c->K);
    return PMX_OK;
}

extern "C" int pmx_prox_apply(pmx_ctx* c, int buf, const pmx_proxseq* prox, const float* step_k) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    if (!c || !prox || !step_k) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    int rc = check_prox(*prox, "prox_apply", true);
    if (rc != PMX_OK) return rc;
    float** slot; int64_t n;
    rc = buf_lookup(c, buf, &slot, &n, false);
    if (rc != PMX_OK) return rc;
    rc = apply_prox_standalone(*slot, n / c->K, (int)c->K, *prox, step_k, c->colpart, c->stream);
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_prox_array(int device, float* X, int64_t rows, int K, const pmx_proxseq* prox, const float* step_k) {
    if (!X || !prox || !step_k) FAIL(PMX_E_INVALID, "NULL argument");
    if (rows <= 0 || K <= 0 || K > MAXK) FAIL(PMX_E_INVALID, "bad shape rows=%lld K=%d", (long long)rows, K);
    int rc = check_prox(*prox, "prox_array", true);
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipSetDevice(device));
    float* d = nullptr;
    const size_t bytes = (size_t)rows * K * sizeof(float), scratch = (size_t)2 * EW_BLOCKS * MAXK * sizeof(double);
    HIP_CHECK(hipMalloc((void**)&d, bytes + scratch + 16));
    double* colpart = reinterpret_cast<double*>(reinterpret_cast<char*>(d) + ((bytes + 15) / 16) * 16);
    hipError_t e = hipMemcpy(d, X, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = apply_prox_standalone(d, rows, K, *prox, step_k, colpart, nullptr);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(X, d, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) FAIL(PMX_E_HIP, "prox_array: %s", hipGetErrorString(e));
    return PMX_OK;
}

// [r6] scratch of pmx_bb_sums, one per device, kept between calls (the stepper calls this once per block and iteration: a hipMalloc / hipFree
// pair each time cost more than the reduction); grows on demand, freed at process exit by the runtime
struct BbScratch { char* d = nullptr; size_t bytes = 0; };
static std::mutex g_bb_mu;
static BbScratch g_bb_scratch[16];
extern "C" int pmx_bb_sums(int device, int is_f64, const void* X, const void* Xprev, const void* G, const void* Gprev, int64_t count, double out[6]) {
    if (!X || !G || !out) FAIL(PMX_E_INVALID, "NULL argument");
    if (count <= 0) FAIL(PMX_E_INVALID, "bad count %lld", (long long)count);
    if ((Xprev == nullptr) != (Gprev == nullptr)) FAIL(PMX_E_INVALID, "X_prev and G_prev come together");
    if (device < 0 || device >= 16) FAIL(PMX_E_INVALID, "bad device %d", device);
    HIP_CHECK(hipSetDevice(device));
    const size_t es = is_f64 ? sizeof(double) : sizeof(float), bytes = ((size_t)count * es + 15) / 16 * 16;
    const int narr = Xprev ? 4 : 2;
    const size_t need = narr * bytes + (BBS_BLOCKS * 6 + 6) * sizeof(double);
    std::lock_guard<std::mutex> lock(g_bb_mu);           // (also serialises the null-stream work of concurrent callers on the one scratch)
    BbScratch& sc = g_bb_scratch[device];
    if (sc.bytes < need) {
        if (sc.d) (void)hipFree(sc.d);
        sc.d = nullptr;
        sc.bytes = 0;
        HIP_CHECK(hipMalloc((void**)&sc.d, need));
        sc.bytes = need;
    }
    char* d = sc.d;
    const void* src[4] = {X, G, Xprev, Gprev};
    hipError_t e = hipSuccess;
    for (int i = 0; i < narr && e == hipSuccess; ++i) e = hipMemcpy(d + i * bytes, src[i], (size_t)count * es, hipMemcpyHostToDevice);
    double* part = reinterpret_cast<double*>(d + narr * bytes);
    if (e == hipSuccess) {
        const char* xp = Xprev ? d + 2 * bytes : nullptr;
        const char* gp = Xprev ? d + 3 * bytes : nullptr;
        if (is_f64) hipLaunchKernelGGL(k_bb_sums<double>, dim3(BBS_BLOCKS), dim3(256), 0, nullptr, (const double*)d, (const double*)xp, (const double*)(d + bytes), (const double*)gp, count, part);
        else hipLaunchKernelGGL(k_bb_sums<float>, dim3(BBS_BLOCKS), dim3(256), 0, nullptr, (const float*)d, (const float*)xp, (const float*)(d + bytes), (const float*)gp, count, part);
        hipLaunchKernelGGL(k_bb_sums_fold, dim3(1), dim3(64), 0, nullptr, part, part + BBS_BLOCKS * 6);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(out, part + BBS_BLOCKS * 6, 6 * sizeof(double), hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) FAIL(PMX_E_HIP, "bb_sums: %s", hipGetErrorString(e));
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
// PGM / FISTA                                             (proxmin/algorithms.py:12-144)
// ------------------------------------------------------------------------------------------------
extern "C" int pmx_pgm_begin(pmx_ctx* c, const pmx_pgm_params* p) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; c->bt_grad_fresh = false; }
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (!p) FAIL(PMX_E_INVALID, "params is NULL");
    for (int j = 0; j < 2; ++j) {
        rc = check_prox(p->prox[j], j ? "prox_S" : "prox_A");
        if (rc != PMX_OK) return rc;
    }
    if (c->f64) {                                // PMX_MODE_F64: plain pgm / FISTA with device operators and a device or fixed step
        if (c->Wd_on && !p->use_fixed_steps && !p->unweighted_rule)     // nmf.step_pgm with an array W raises (nmf.py:63)
            FAIL(PMX_E_INVALID, "The truth value of an array with more than one element is ambiguous. Use a.any() or a.all()");
        if ((p->backtracking && !c->f64big) || p->bb_type || p->host_prox[0] || p->host_prox[1])
            FAIL(PMX_E_UNSUPPORTED, "fp64 contexts run pgm / FISTA with this library's operators and step rules (no Barzilai-Borwein or user prox; the line search on the matrix-core kernels only: PMX_MODE_F64_MFMA)");
        c->pgm = *p;
        c->algo = ALG_PGM;
        c->it = 0;
        c->nest_t = 1.0;
        rc = reset_status(c);
        if (rc != PMX_OK) return rc;
        c->btT[0] = c->btT[1] = 1.0;
        if (p->accelerated || p->backtracking) {      // (the line search keeps the evaluation point apart from the iterate: algorithms.py:96-97)
            for (int j = 0; j < 2; ++j) {
                rc = dallocT(c, &c->Xed[j], (size_t)c->rows[j] * c->K, false);
                if (rc == PMX_OK && p->backtracking) rc = dallocT(c, &c->Xprevd[j], (size_t)c->rows[j] * c->K, false);
                if (rc != PMX_OK) return rc;
                HIP_CHECK(hipMemcpyAsync(c->Xed[j], c->Xd[j], c->rows[j] * c->K * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            }
        }
        if (p->accelerated) {
            c->nest_t = 0.5 * (1.0 + sqrt(4.0 * c->nest_t * c->nest_t + 1.0));       // the first omega (== 0) is consumed at it = 0
        }
        return PMX_OK;
    }
    if (c->W && !p->use_fixed_steps && !p->bb_type && !p->unweighted_rule)   // nmf.step_pgm with an array W raises (nmf.py:63)
        FAIL(PMX_E_INVALID, "The truth value of an array with more than one element is ambiguous. Use a.any() or a.all()");
    rc = dallocT(c, &c->tickets, 4);
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipMemsetAsync(c->tickets, 0, sizeof(unsigned), c->stream));   // (launches skipped by a halted chain took no tickets)
    c->ticketLaunches = 0;
    c->pgm = *p;
    c->algo = ALG_PGM;
    c->it = 0;
    c->decide_pending = false;
    c->nest_t = 1.0;
    c->omega_cur = 0.f;
    rc = reset_status(c);
    if (rc != PMX_OK) return rc;
    if (p->bb_type != 0 && p->bb_type != 1 && p->bb_type != 2) FAIL(PMX_E_INVALID, "bb_type must be 0, 1 or 2");   // utils.py:212
    if (p->bb_type && p->backtracking) FAIL(PMX_E_UNSUPPORTED, "Barzilai-Borwein steps together with backtracking are not implemented");
    if (p->bb_type)
        for (int j = 0; j < 2; ++j) {
            rc = dallocT(c, &c->bbX[j], (size_t)c->rows[j] * c->K, false);
            if (rc == PMX_OK) rc = dallocT(c, &c->bbG[j], (size_t)c->rows[j] * c->K, false);
            if (rc != PMX_OK) return rc;
        }
    c->bt_pending = 0;
    if (p->backtracking)
        for (int j = 0; j < 2; ++j)
            if (p->host_prox[j]) {               // [r4] every trial of that block takes a host round trip (pmx_pgm_bt_split)
                rc = dallocT(c, &c->btBuf[j], (size_t)c->rowsK[j] * c->K, c->framed);
                if (rc != PMX_OK) return rc;
            }
    for (int j = 0; j < 2; ++j)
        if (p->host_prox[j]) {
            rc = dallocT(c, &c->Xp[j], (size_t)c->rowsK[j] * c->K, c->framed);
            if (rc != PMX_OK) return rc;
        }
    c->btT[0] = c->btT[1] = 1.0;
    if (p->backtracking) {   // host-driven trials read device sums after every K1 pass: no room for a repeated iteration
        rc = one_iteration_per_call(c);
        if (rc != PMX_OK) return rc;
    }
    if (p->backtracking)
        for (int j = 0; j < 2; ++j) {
            rc = dallocT(c, &c->Xp[j], (size_t)c->rowsK[j] * c->K, c->framed);
            if (rc == PMX_OK) rc = dallocT(c, &c->Xe[j], (size_t)c->rowsK[j] * c->K, c->framed);
            if (rc != PMX_OK) return rc;
            HIP_CHECK(hipMemcpyAsync(c->Xe[j], c->X[j], c->rows[j] * c->K * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
    if (p->accelerated) {
        for (int j = 0; j < 2; ++j) {
            rc = dallocT(c, &c->Xe[j], (size_t)c->rowsK[j] * c->K, c->framed);
            if (rc != PMX_OK) return rc;
            // omega = 0 on the first read (utils.py:201-203): the first extrapolated point is X itself
            HIP_CHECK(hipMemcpyAsync(c->Xe[j], c->X[j], c->rows[j] * c->K * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        // consume the first omega (== 0) like `accel.omega` at it = 0
        const double t1 = 0.5 * (1.0 + sqrt(4.0 * c->nest_t * c->nest_t + 1.0));
        c->nest_t = t1;
    }
    return PMX_OK;
}

// omega that the NEXT iteration will read (utils.py:198-206)
static float next_omega(pmx_ctx* c) {
    if (!c->pgm.accelerated) return 0.f;
    const double t = c->nest_t;
    const double t1 = 0.5 * (1.0 + sqrt(4.0 * t * t + 1.0));
    const double om = (t - 1.0) / t1;
    c->nest_t = t1;
    return (float)om;
}

static int pgm_enqueue_iteration(pmx_ctx* c) {
    const pmx_pgm_params& p = c->pgm;
    const float* A = p.accelerated ? c->Xe[0] : c->X[0];
    const float* St = p.accelerated ? c->Xe[1] : c->X[1];
    int rc;
    if (c->use_small && eig_small_applies(c) && !p.use_fixed_steps && !p.bb_type && !(getenv("PMX_SMALL_FRONT") && atoi(getenv("PMX_SMALL_FRONT")) == 0)) {
        rc = enqueue_small_front(c, A, St, (double)p.step_scale);          // algorithms.py:105-106, one launch
        if (rc != PMX_OK) return rc;
    } else {
        // [r6] Where the partial Gram matrices of this iterate were left by the previous update kernel and K1 is one of the producer / consumer kernels of
        // K = 32 (cfg2's: k_grad_f16_k32, k_grad_f32_pc<32>), their fold (and the previous iteration's stopping test) ride in K1's first workgroups and
        // k_eig follows K1: three launches per iteration instead of four (k_gram_reduce was 5.5 us of cfg2's 50; pmx_common.h: k1_gram_fold).
        const bool fold_here = !p.use_fixed_steps && !p.bb_type && c->gram_by_update && c->fold_in_k1 && !c->host_grad && !eig_small_applies(c) && c->KP == 32 && c->Kk == 32 &&
                               (c->k32f16 || (c->f32pc && !c->use_small && !c->use_bf16)) && !c->bsd_decide_pending &&
                               c->plan.gridX * c->plan.gridY >= k1_gram_fold_wgs(c->KP, 512);
        if (fold_here) {
            K1GramFold gf{};
            gf.part = c->gramPart; gf.G = c->gramG; gf.KP = c->KP;
            gf.nparts[0] = gram_nparts(c->M); gf.nparts[1] = gram_nparts(c->N);
            if (c->decide_pending) { gf.dec_partials = c->partials; gf.dec_status = c->dstatus; gf.dec_e_rel[0] = p.e_rel[0]; gf.dec_e_rel[1] = p.e_rel[1]; }
            c->k1_fold = gf;
            c->decide_pending = false;
        } else if (!p.use_fixed_steps && !p.bb_type) {
            rc = enqueue_steps(c, A, St, true, true, (double)p.step_scale, c->gram_by_update, c->decide_pending);   // algorithms.py:106
            if (rc != PMX_OK) return rc;
            c->decide_pending = false;
        }
        rc = enqueue_grad(c, A, St, 1, 1, c->absmax_by_finish);               // algorithms.py:105
        if (rc != PMX_OK) return rc;
        if (fold_here) {
            if (c->k1_fold.part != nullptr) {    // the launch that ran is not one that carries the fold (a fall-back inside enqueue_grad): the stand-alone kernel, behind K1
                GramReduceArgs r{};
                r.part = c->k1_fold.part; r.G = c->gramG; r.KP = c->KP; r.status = c->dstatus;
                r.want[0] = r.want[1] = 1;
                r.nparts[0] = c->k1_fold.nparts[0]; r.nparts[1] = c->k1_fold.nparts[1];
                r.dec_partials = c->k1_fold.dec_partials; r.dec_status = c->k1_fold.dec_status;
                r.dec_e_rel[0] = c->k1_fold.dec_e_rel[0]; r.dec_e_rel[1] = c->k1_fold.dec_e_rel[1];
                c->k1_fold = K1GramFold{};
                launch_gram_reduce(r, c->stream);
            }
            rc = enqueue_eig_only(c, true, true, (double)p.step_scale);     // algorithms.py:106, behind K1: the update kernel is its only reader
            if (rc != PMX_OK) return rc;
        }
    }
    if (p.bb_type) {                                                      // step(*_X, it, grads=G): utils.py:216-241
        BBArgs b{};
        b.X[0] = A; b.X[1] = St;
        for (int j = 0; j < 2; ++j) {
            b.slab[j] = slab_ref(c, j);
            b.G[j] = c->G[j]; b.Xprev[j] = c->bbX[j]; b.Gprev[j] = c->bbG[j];
            b.rows[j] = c->rows[j];
        }
        b.K = (int)c->K; b.status = c->dstatus; b.partials = c->partials;
        b.first = c->it == 0;
        launch_bb_reduce(b, c->stream);
        BBStepArgs bs{};
        bs.status = c->dstatus; bs.partials = c->partials; bs.it = c->it; bs.type = p.bb_type; bs.init_r = p.bb_init_r;
        launch_bb_step(bs, c->stream);
    }
    PgmArgs u{};
    for (int j = 0; j < 2; ++j) {
        u.X[j] = c->X[j];
        u.Xe[j] = p.accelerated ? c->Xe[j] : c->X[j];
        u.G[j] = c->G[j];
        u.slab[j] = slab_ref(c, j);
        if (p.bb_type) { u.slab[j].base = c->G[j]; u.slab[j].n = 1; u.slab[j].ld = (int)c->K; u.slab[j].extra = nullptr; }    // already folded by k_bb_reduce
        u.rows[j] = c->rows[j];
        u.prox[j] = to_dev(p.prox[j]);
    }
    u.K = (int)c->K;
    u.status = c->dstatus;
    u.partials = c->partials;
    u.accelerated = p.accelerated;
    u.omega_next = next_omega(c);
    // the stopping test (algorithms.py:130-135) is made by the last of the update kernel's 2 x EW_BLOCKS workgroups
    if (c->ticketLaunches >= (1u << 30)) {
        HIP_CHECK(hipMemsetAsync(c->tickets, 0, sizeof(unsigned), c->stream));
        c->ticketLaunches = 0;
    }
    // [r4] where the next launch on the stream is the step rule's k_gram_reduce, the stopping test rides in IT (beside the fold)
    // instead of in this kernel's last-arriving workgroup (behind everybody else): pgm_flush_decide() covers the last iteration of a chunk
    const bool defer_decide = !p.use_fixed_steps && !p.bb_type && !eig_small_applies(c);     // (small factors: k_eig_small, no k_gram_reduce launch)
    u.tickets = defer_decide ? nullptr : c->tickets;
    {
        const int64_t rmax = c->rows[0] > c->rows[1] ? c->rows[0] : c->rows[1];
        u.nbx = rmax <= (int64_t)EW_BLOCKS * (EW_THREADS / 32) ? (int)((rmax + EW_THREADS / 32 - 1) / (EW_THREADS / 32)) : EW_BLOCKS;
        if (!defer_decide) {
            c->ticketLaunches += 2u * (unsigned)u.nbx;   // tickets drawn so far
            u.ticket_last = c->ticketLaunches - 1u;
        }
    }
    // the fp16 K1's operand maxima for the next iteration come from this kernel (every workgroup writes its partial: full grid only)
    u.absmax_out = c->f16_scales ? c->absmax : nullptr;       // ([r4] any grid: the workgroups that exist zero the slots of those that do not)
    u.e_rel[0] = p.e_rel[0]; u.e_rel[1] = p.e_rel[1];
    // the next iteration's partial Gram matrices from this launch (PgmArgs::gramPart): the Lipschitz rule on factors of <= 4096
    // rows each, K <= 64 (cfg2); PMX_GRAM_IN_UPDATE=0 keeps k_gram_partial (A/B)
    const bool gram_here = !p.use_fixed_steps && !p.bb_type && !c->use_small && c->K <= 64 && c->rows[0] <= 4096 && c->rows[1] <= 4096 && c->gram_in_update;
    u.gramPart = gram_here ? c->gramPart : nullptr;
    u.KP = c->KP;
    launch_pgm_update(u, c->stream);                                      // algorithms.py:107-108
    HIP_CHECK(hipGetLastError());
    c->gram_by_update = gram_here;
    c->decide_pending = defer_decide;
    c->absmax_by_finish = u.absmax_out != nullptr;
    c->it += 1;
    return PMX_OK;
}

static int loss_now(pmx_ctx* c, const float* A, const float* St, double* out) {
    int rc = enqueue_grad(c, A, St, 0, 0);
    if (rc != PMX_OK) return rc;
    const int n = c->nloss;
    std::vector<double> h(n);
    HIP_CHECK(hipMemcpyAsync(h.data(), c->lossPart, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += h[i];
    *out = 0.5 * s;
    return PMX_OK;
}

// one PGM iteration with the Beck-Teboulle line search (algorithms.py:93-135); host-driven, synchronous.
// [r4] Resumable: a block whose prox is a user callable (pmx_pgm_params::host_prox) has every TRIAL of the search take a host
// round trip -- the device forms the callable's argument T_j = _X_j - T_j s_j G_j (k_bt_update mode 1, PMX_BUF_BT_A / _ST),
// returns with *need = the blocks waiting for their prox (and eff[j] = T_j s_j, the step the reference passes it, :108,125),
// and the next call (phase 1) adopts what the caller left there (mode 2) and goes on with the test.  *need == 0: the
// iteration is complete.  Blocks with this library's operators never leave the device.
static BtArgs bt_args(pmx_ctx* c) {
    const pmx_pgm_params& p = c->pgm;
    BtArgs u{};
    for (int j = 0; j < 2; ++j) {
        u.X[j] = c->X[j]; u.Xe[j] = c->Xe[j]; u.Xp[j] = c->Xp[j]; u.G[j] = c->G[j];
        u.slab[j] = slab_ref(c, j);
        u.rows[j] = c->rows[j];
        u.prox[j] = to_dev(p.prox[j]);
        u.T[j] = (float)c->btT[j];
        u.Tb[j] = c->btBuf[j];
    }
    u.K = (int)c->K; u.status = c->dstatus; u.partials = c->partials;
    return u;
}
static int bt_step(pmx_ctx* c, int phase, int* need, double eff[2]) {
    const pmx_pgm_params& p = c->pgm;
    int rc;
    *need = 0;
    BtCollectArgs col{};
    col.status = c->dstatus; col.partials = c->partials;
    if (phase == 0) {
        if (c->bt_pending) FAIL(PMX_E_STATE, "the line search is waiting for the prox of block mask %d (phase 1)", c->bt_pending);
        // _X is a separate buffer in this mode (algorithms.py:96-97): Xe already holds it (copy of X, or the
        // extrapolated point written by k_bt_finish at the end of the previous iteration)
        if (c->it == 0) {                                        // f_prev = f(*X_) on the first iteration (:113-114)
            rc = loss_now(c, c->X[0], c->X[1], &c->bt_fprev);
            if (rc != PMX_OK) return rc;
        }
        if (!p.use_fixed_steps && !p.bb_type) {
            rc = enqueue_steps(c, c->Xe[0], c->Xe[1], true, true, (double)p.step_scale);
            if (rc != PMX_OK) return rc;
        }
        if (!c->bt_grad_fresh) {
            rc = enqueue_grad(c, c->Xe[0], c->Xe[1], 1, 1);
            if (rc != PMX_OK) return rc;
        }
        c->bt_grad_fresh = false;
        BtArgs u = bt_args(c);
        u.first = 1;
        for (int j = 0; j < 2; ++j) { u.do_block[j] = 1; u.mode[j] = p.host_prox[j] ? 1 : 0; }
        launch_bt_update(u, c->stream);
        c->bt_pending = (p.host_prox[0] ? 1 : 0) | (p.host_prox[1] ? 2 : 0);
        c->bt_trial = 3;                                         // blocks whose sums this trial renews: both
        if (c->bt_pending) {
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            for (int j = 0; j < 2; ++j) eff[j] = c->btT[j] * c->hstatus->step[j];
            *need = c->bt_pending;
            return PMX_OK;
        }
    } else {
        if (!c->bt_pending) FAIL(PMX_E_STATE, "no block of the line search is waiting for its prox");
        BtArgs u = bt_args(c);
        u.first = 0;
        for (int j = 0; j < 2; ++j) { u.do_block[j] = (c->bt_pending >> j) & 1; u.mode[j] = 2; }
        launch_bt_update(u, c->stream);
        c->bt_pending = 0;
    }
    for (int guard = 0; guard < 200; ++guard) {
        col.do_block[0] = c->bt_trial & 1; col.do_block[1] = (c->bt_trial >> 1) & 1;
        launch_bt_collect(col, c->stream);
        double f_now;
        rc = loss_now(c, c->X[0], c->X[1], &f_now);
        if (rc != PMX_OK) return rc;
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        const DevStatus* s = c->hstatus;
        double q = 0.0;
        for (int j = 0; j < 2; ++j) q += s->bt[j][0] + 0.5 / (c->btT[j] * s->step[j]) * s->bt[j][1];
        if (!(f_now > c->bt_fprev + q)) { c->bt_fnow = f_now; break; }   // algorithms.py:117-118
        // block with the largest relative update direction (algorithms.py:121)
        const double r0 = s->step[0] * s->bt[0][2] / s->bt[0][3], r1 = s->step[1] * s->bt[1][2] / s->bt[1][3];
        const int jm = r1 > r0 ? 1 : 0;                          // np.argmax: first maximum
        c->btT[jm] *= 0.5;
        BtArgs u = bt_args(c);
        u.first = 0;
        u.do_block[0] = jm == 0; u.do_block[1] = jm == 1;
        u.mode[jm] = p.host_prox[jm] ? 1 : 0;
        launch_bt_update(u, c->stream);
        c->bt_trial = 1 << jm;
        c->bt_fnow = f_now;
        if (p.host_prox[jm]) {
            c->bt_pending = 1 << jm;
            for (int j = 0; j < 2; ++j) eff[j] = c->btT[j] * s->step[j];
            *need = c->bt_pending;
            return PMX_OK;
        }
    }
    c->bt_fprev = c->bt_fnow;                                    // :127
    BtFinishArgs fin{};
    for (int j = 0; j < 2; ++j) { fin.X[j] = c->X[j]; fin.Xp[j] = c->Xp[j]; fin.Xe[j] = c->Xe[j]; fin.rows[j] = c->rows[j]; }
    fin.K = (int)c->K; fin.status = c->dstatus;
    fin.omega_next = next_omega(c);
    launch_bt_finish(fin, c->stream);
    DecideArgs d{};
    d.status = c->dstatus; d.partials = c->partials;
    d.e_rel[0] = p.e_rel[0]; d.e_rel[1] = p.e_rel[1];
    d.check = 1;
    launch_pgm_decide(d, c->stream);
    HIP_CHECK(hipGetLastError());
    c->it += 1;
    return PMX_OK;
}
static int pgm_bt_iteration(pmx_ctx* c) {                        // every operator on the device: never waits for the caller
    int need = 0;
    double eff[2];
    int rc = bt_step(c, 0, &need, eff);
    if (rc == PMX_OK && need) FAIL(PMX_E_STATE, "a user-defined prox inside the line search runs through pmx_pgm_bt_split");
    return rc;
}

static int set_fixed_steps(pmx_ctx* c, const double s[2]) {
    HIP_CHECK(hipMemcpyAsync(&c->dstatus->step[0], s, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return PMX_OK;
}

static int pgm_flush_decide(pmx_ctx* c) {
    if (!c->decide_pending) return PMX_OK;
    DecideArgs d{};
    d.status = c->dstatus; d.partials = c->partials;
    d.e_rel[0] = c->pgm.e_rel[0]; d.e_rel[1] = c->pgm.e_rel[1];
    d.check = 1;
    launch_pgm_decide(d, c->stream);
    HIP_CHECK(hipGetLastError());
    c->decide_pending = false;
    return PMX_OK;
}

extern "C" int pmx_pgm_run(pmx_ctx* c, int n_iter, pmx_result* res) {
    if (c) c->absmax_by_finish = false;      // (gram_by_update survives: nothing but this solver's own update kernel has touched the factors since)
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_PGM) FAIL(PMX_E_STATE, "pmx_pgm_begin has not been called");
    if (n_iter < 0) FAIL(PMX_E_INVALID, "n_iter < 0");
    const int it0 = c->hstatus->it_done;
    if (c->pgm.use_fixed_steps) {
        rc = set_fixed_steps(c, c->pgm.fixed_steps);
        if (rc != PMX_OK) return rc;
    }
    if (c->f64 && c->pgm.backtracking) {     // [r6] host-driven trials: one iteration at a time
        for (int i = 0; i < n_iter && !c->hstatus->stopped; ++i) {
            rc = pgm64_bt_iteration(c);
            if (rc != PMX_OK) return rc;
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
        }
        fill_result(c, res, it0);
        return PMX_OK;
    }
    if (c->f64) {
        for (int left = n_iter; left > 0 && !c->hstatus->stopped; left -= 32) {
            for (int i = 0; i < std::min(left, 32); ++i) {
                rc = pgm64_enqueue_iteration(c);
                if (rc != PMX_OK) return rc;
            }
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            if (c->hstatus->halt && c->hstatus->reason == HALT_ERROR) FAIL(PMX_E_HIP, "%s", chain_error_text(c));
        }
        fill_result(c, res, it0);
        return PMX_OK;
    }
    int left = n_iter;
    while (left > 0 && !c->hstatus->stopped) {
        const int chunk = c->pgm.backtracking ? 1 : std::min(left, 32);
        for (int i = 0; i < chunk; ++i) {
            rc = c->pgm.backtracking ? pgm_bt_iteration(c) : pgm_enqueue_iteration(c);
            if (rc != PMX_OK) return rc;
        }
        rc = pgm_flush_decide(c);
        if (rc != PMX_OK) return rc;
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        int again = 0;
        rc = chain_fault_fallback(c, &again);
        if (rc != PMX_OK) return rc;
        if (again) {                      // iterations after the faulting one were skipped: redo them
            c->it = c->hstatus->it_done;
            if (c->pgm.accelerated) {     // the host-side Nesterov sequence ran ahead with the skipped iterations
                c->nest_t = 1.0;
                for (int i = 0; i <= c->it; ++i) c->nest_t = 0.5 * (1.0 + sqrt(4.0 * c->nest_t * c->nest_t + 1.0));
            }
            left = n_iter - (c->hstatus->it_done - it0);
            continue;
        }
        left -= chunk;
    }
    fill_result(c, res, it0);
    return PMX_OK;
}

extern "C" int pmx_pgm_set_fixed_steps(pmx_ctx* c, const double steps[2]) {
    if (!c || !steps) FAIL(PMX_E_INVALID, "NULL argument");
    if (c->algo != ALG_PGM) FAIL(PMX_E_STATE, "pmx_pgm_begin has not been called");
    if (!c->pgm.use_fixed_steps) FAIL(PMX_E_STATE, "the context was not begun with fixed steps");
    HIP_CHECK(hipSetDevice(c->device));
    c->pgm.fixed_steps[0] = steps[0];
    c->pgm.fixed_steps[1] = steps[1];
    return set_fixed_steps(c, c->pgm.fixed_steps);    // (the context's own copy: the asynchronous upload outlives the caller's array)
}

extern "C" int pmx_pgm_step_arrays(pmx_ctx* c, int mask) {
    REJECT_F64(c);
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (mask < 0 || mask > 3) FAIL(PMX_E_INVALID, "mask must be 0..3");
    for (int j = 0; j < 2; ++j)
        if (((mask >> j) & 1) && !c->stepArr[j]) FAIL(PMX_E_STATE, "block %d: upload the steps into PMX_BUF_STEP_%s first", j, j ? "ST" : "A");
    c->step_arr_mask = mask;
    return PMX_OK;
}

extern "C" int pmx_pgm_bt_split(pmx_ctx* c, int phase, int* need, double eff_steps[2], pmx_result* res) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (!need || !eff_steps) FAIL(PMX_E_INVALID, "NULL argument");
    if (c->algo != ALG_PGM || !c->pgm.backtracking) FAIL(PMX_E_STATE, "pmx_pgm_begin has not been called with backtracking");
    if (phase != 0 && phase != 1) FAIL(PMX_E_INVALID, "bad phase %d", phase);
    const int it0 = c->hstatus->it_done;
    if (phase == 0 && c->pgm.use_fixed_steps) {
        rc = set_fixed_steps(c, c->pgm.fixed_steps);
        if (rc != PMX_OK) return rc;
    }
    { rc = one_iteration_per_call(c); if (rc != PMX_OK) return rc; }       // (one iteration per call: nothing to repeat into)
    rc = bt_step(c, phase, need, eff_steps);
    if (rc != PMX_OK) return rc;
    if (*need == 0) {
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        fill_result(c, res, it0);
    }
    return PMX_OK;
}

extern "C" int pmx_pgm_split(pmx_ctx* c, int phase, const double* steps, pmx_result* res) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_PGM) FAIL(PMX_E_STATE, "pmx_pgm_begin has not been called");
    const pmx_pgm_params& p = c->pgm;
    // (phase 0 alone -- the gradient at the evaluation point, for a user `step` that wants `grads` -- is harmless next to
    //  the line search: the iteration itself then runs through pmx_pgm_run(ctx, 1) with the steps of pmx_pgm_set_fixed_steps)
    if (p.backtracking && phase != 0) FAIL(PMX_E_UNSUPPORTED, "pmx_pgm_split: not with backtracking");
    const float* A = p.accelerated ? c->Xe[0] : c->X[0];
    const float* St = p.accelerated ? c->Xe[1] : c->X[1];
    const bool any_host = p.host_prox[0] || p.host_prox[1];
    auto update = [&](int stage) {       // stage 1: "pre" of the host blocks; stage 2: the update
        PgmArgs u{};
        for (int j = 0; j < 2; ++j) {
            u.X[j] = c->X[j];
            u.Xe[j] = p.accelerated ? c->Xe[j] : c->X[j];
            u.G[j] = c->G[j];
            u.slab[j].base = c->G[j];    // folded by phase 0
            u.slab[j].n = 1; u.slab[j].ld = (int)c->K;
            u.rows[j] = c->rows[j];
            u.prox[j] = to_dev(p.prox[j]);
            u.T[j] = c->Xp[j];
            u.stepArr[j] = (c->step_arr_mask >> j) & 1 ? c->stepArr[j] : nullptr;
            u.mode[j] = stage == 1 ? (p.host_prox[j] ? 1 : 3) : (p.host_prox[j] ? 2 : 0);
        }
        u.K = (int)c->K;
        u.status = c->dstatus;
        u.partials = c->partials;
        u.accelerated = p.accelerated;
        u.omega_next = stage == 2 ? next_omega(c) : 0.f;
        launch_pgm_update(u, c->stream);
    };
    switch (phase) {
        case 0: {
            { rc = one_iteration_per_call(c); if (rc != PMX_OK) return rc; }   // (one iteration per call: nothing to repeat into)
            if (!p.use_fixed_steps && !p.bb_type) {
                rc = enqueue_steps(c, A, St, true, true, (double)p.step_scale);
                if (rc != PMX_OK) return rc;
            } else if (p.use_fixed_steps && !steps) {
                // nmf.constant_step with a host-side prox: reset_status left DevStatus::step at 0 and only pmx_pgm_run
                // uploads the constants -- without them the host prox would be handed T = Xe - 0 G
                rc = set_fixed_steps(c, p.fixed_steps);
                if (rc != PMX_OK) return rc;
            }
            rc = enqueue_grad(c, A, St, 1, 1);
            if (rc != PMX_OK) return rc;
            c->bt_grad_fresh = p.backtracking != 0;      // (the line search's phase 0 evaluates the same point: Xe)
            if (p.bb_type) {             // the Barzilai-Borwein rule on the device (utils.py:216-241), exactly as a fused iteration
                BBArgs b{};              // does it: k_bb_reduce folds the gradient as well
                b.X[0] = A; b.X[1] = St;
                for (int j = 0; j < 2; ++j) {
                    b.slab[j] = slab_ref(c, j);
                    b.G[j] = c->G[j]; b.Xprev[j] = c->bbX[j]; b.Gprev[j] = c->bbG[j];
                    b.rows[j] = c->rows[j];
                }
                b.K = (int)c->K; b.status = c->dstatus; b.partials = c->partials;
                b.first = c->it == 0;
                launch_bb_reduce(b, c->stream);
                BBStepArgs bs{};
                bs.status = c->dstatus; bs.partials = c->partials; bs.it = c->it; bs.type = p.bb_type; bs.init_r = p.bb_init_r;
                launch_bb_step(bs, c->stream);
            } else {
                FoldArgs f{};
                for (int j = 0; j < 2; ++j) { f.slab[j] = slab_ref(c, j); f.G[j] = c->G[j]; f.rows[j] = c->rows[j]; }
                f.K = (int)c->K;
                f.status = c->dstatus;
                launch_fold(f, 2, c->stream);
            }
            HIP_CHECK(hipGetLastError());
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            if (res) { res->steps[0] = c->hstatus->step[0]; res->steps[1] = c->hstatus->step[1]; }
            return PMX_OK;
        }
        case 1:
            if (!any_host) return PMX_OK;
            if (steps) { rc = set_fixed_steps(c, steps); if (rc != PMX_OK) return rc; }
            update(1);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(c->stream));
            return PMX_OK;
        case 2: {
            const int it0 = c->hstatus->it_done;
            if (steps) { rc = set_fixed_steps(c, steps); if (rc != PMX_OK) return rc; }
            update(2);
            DecideArgs d{};
            d.status = c->dstatus;
            d.partials = c->partials;
            d.e_rel[0] = p.e_rel[0]; d.e_rel[1] = p.e_rel[1];
            d.check = 1;
            launch_pgm_decide(d, c->stream);
            HIP_CHECK(hipGetLastError());
            c->it += 1;
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            fill_result(c, res, it0);
            return PMX_OK;
        }
        default: FAIL(PMX_E_INVALID, "bad phase %d", phase);
    }
}

// ------------------------------------------------------------------------------------------------
// adaprox                                                 (proxmin/algorithms.py:248-423)
// ------------------------------------------------------------------------------------------------
extern "C" int pmx_adaprox_begin(pmx_ctx* c, const pmx_adaprox_params* p, int warm_moments) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (!p) FAIL(PMX_E_INVALID, "params is NULL");
    if (p->scheme < PMX_ADAM || p->scheme > PMX_RADAM) FAIL(PMX_E_INVALID, "unknown scheme %d", p->scheme);
    if (!(p->b2 >= 0 && p->b2 < 1)) FAIL(PMX_E_INVALID, "b2 out of [0,1)");           // algorithms.py:332
    if (!(p->eps >= 0)) FAIL(PMX_E_INVALID, "eps < 0");                               // :333
    if (!(p->p > 0 && p->p <= 0.5)) FAIL(PMX_E_INVALID, "p out of (0,0.5]");           // :334
    for (int j = 0; j < 2; ++j) {
        rc = check_prox(p->prox[j], j ? "prox_S" : "prox_A");
        if (rc != PMX_OK) return rc;
    }
    if (c->f64) {                                // PMX_MODE_F64: this library's operators and step rule (or two constants), k64_ada_iter
        if (p->host_prox[0] || p->host_prox[1] || p->use_fixed_steps == 2)
            FAIL(PMX_E_UNSUPPORTED, "fp64 contexts run adaprox with this library's operators and step rule (no user prox / step)");
        c->ada = *p;
        c->algo = ALG_ADAPROX;
        c->it = 0;
        rc = reset_status(c);
        if (rc != PMX_OK) return rc;
        for (int j = 0; j < 2; ++j) {
            const size_t n = (size_t)c->rows[j] * c->K;
            const bool fresh_m = c->Md[j] == nullptr, fresh_v = c->Vd[j] == nullptr;
            rc = dallocT(c, &c->Md[j], n);
            if (rc == PMX_OK) rc = dallocT(c, &c->Vd[j], n);
            if (rc != PMX_OK) return rc;
            if (!warm_moments) {   // cold start: zeros (algorithms.py:348-353)
                if (!fresh_m) HIP_CHECK(hipMemsetAsync(c->Md[j], 0, n * sizeof(double), c->stream));
                if (!fresh_v) HIP_CHECK(hipMemsetAsync(c->Vd[j], 0, n * sizeof(double), c->stream));
            }
            if (p->warm_vhat && !c->Vhd[j]) FAIL(PMX_E_STATE, "warm_vhat set but Vhat buffers were not uploaded");
            if (p->check_convergence) rc = dallocT(c, &c->Xpd[j], n, false);
            if (rc == PMX_OK && p->prox[j].n > 0) {
                rc = dallocT(c, &c->Psid[j], n, false);
                if (rc == PMX_OK) rc = dallocT(c, &c->zd[j], n, false);
            }
            if (rc != PMX_OK) return rc;
        }
        rc = dallocT(c, &c->alpha64, c->f64big ? 2 * MAXK : 32);
        if (rc == PMX_OK && c->f64big) rc = dallocT(c, &c->colpart64, (size_t)2 * EW_BLOCKS * MAXK);
        if (rc != PMX_OK) return rc;
        c->nsub64 = 4;
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return PMX_OK;
    }
    c->ada = *p;
    c->algo = ALG_ADAPROX;
    c->it = 0;
    c->nsub_guess = 2;
    c->sub_nt = (getenv("PMX_SUB_BATCH") && atoi(getenv("PMX_SUB_BATCH")) == 1) ? 1 : SUB_NT_MAX;
    for (auto& r : c->sub_rec) r = pmx_ctx::SubRec{};
    rc = reset_status(c);
    if (rc != PMX_OK) return rc;
    for (int j = 0; j < 2; ++j) {
        const size_t n = (size_t)c->rows[j] * c->K;
        const bool fresh_m = c->Mm[j] == nullptr, fresh_v = c->Vv[j] == nullptr;
        rc = dallocT(c, &c->Mm[j], n);
        if (rc == PMX_OK) rc = dallocT(c, &c->Vv[j], n);
        if (rc != PMX_OK) return rc;
        if (!warm_moments) {   // cold start: zeros (algorithms.py:348-353)
            if (!fresh_m) HIP_CHECK(hipMemsetAsync(c->Mm[j], 0, n * sizeof(float), c->stream));
            if (!fresh_v) HIP_CHECK(hipMemsetAsync(c->Vv[j], 0, n * sizeof(float), c->stream));
        }
        if (p->warm_vhat && !c->Vh[j]) FAIL(PMX_E_STATE, "warm_vhat set but Vhat buffers were not uploaded");
        if (p->check_convergence) {
            rc = dallocT(c, &c->Xp[j], n, false);
            if (rc != PMX_OK) return rc;
        }
        if (p->prox[j].n > 0 || p->host_prox[j]) {
            rc = dallocT(c, &c->Psi[j], n, false);
            if (rc == PMX_OK) rc = dallocT(c, &c->zb[j][0], n, false);
            if (rc == PMX_OK) rc = dallocT(c, &c->zb[j][1], n, false);
            if (rc != PMX_OK) return rc;
        }
    }
    // the iteration tail as one persistent kernel where its LDS-resident state fits and one workgroup per CU can be
    // resident (PMX_TAIL_FUSED=0: the four separate kernels; PMX_SUB_BATCH=1 implies them)
    c->tail_fused = false;
    if (!(getenv("PMX_TAIL_FUSED") && atoi(getenv("PMX_TAIL_FUSED")) == 0) && c->sub_nt != 1 && c->tailFaults == 0 &&
        !p->host_prox[0] && !p->host_prox[1]) {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) ncu = 0;
        TailArgs probe{};
        probe.m.K = (int)c->K;
        for (int j = 0; j < 2; ++j) probe.slots[j] = (int)((c->rows[j] + 8191) / 8192);
        if (ncu >= EW_BLOCKS && ada_tail_lds_bytes(probe) <= ADA_TAIL_LDS_MAX) {
            rc = dallocT(c, &c->gridbar, 1);
            if (rc == PMX_OK && getenv("PMX_TAIL_PROF")) rc = dallocT(c, &c->tailprof, 16);
            if (rc != PMX_OK) return rc;
            c->tail_fused = true;
        }
    }
    // step sizes of the first iteration from the initial factors (algorithms.py:370 -> nmf.py:93)
    AlphaArgs al = alpha_args(c);
    al.use_fixed = p->use_fixed_steps;
    al.fixed[0] = (float)p->fixed_alpha[0];
    al.fixed[1] = (float)p->fixed_alpha[1];
    rc = enqueue_alpha_from_factors(c, al);
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

static SubArgs sub_args(pmx_ctx* c, int t) {
    const pmx_adaprox_params& p = c->ada;
    SubArgs s{};
    for (int j = 0; j < 2; ++j) {
        s.X[j] = s_view(c, j, c->X[j]);
        s.Psi[j] = s_view(c, j, c->Psi[j]);
        s.zb[j][0] = s_view(c, j, c->zb[j][0]);
        s.zb[j][1] = s_view(c, j, c->zb[j][1]);
        s.rows[j] = upd_rows(c, j);
        s.prox[j] = to_dev(p.prox[j]);
        s.e_rel[j] = p.e_rel[j];
        s.has_prox[j] = p.prox[j].n > 0;
    }
    s.K = (int)c->K;
    s.status = c->dstatus;
    s.partials = c->partials;
    s.t = t;
    s.nt = c->sub_nt;
    s.prox_max_iter = p.prox_max_iter;
    return s;
}

// enqueue sub-iteration launches covering at least passes [t0, t0 + n); t0 is a multiple of the launch size.
// Returns the number of passes enqueued so far (a multiple of the launch size).
static int ada_enqueue_subs(pmx_ctx* c, int t0, int n) {
    const int nt = c->sub_nt;
    int t = t0;
    for (; t < t0 + n; t += nt) launch_ada_sub(sub_args(c, t), c->stream);
    return t;
}

// tail of an iteration: finish + decide.  t = number of sub-iteration passes enqueued so far
static int ada_enqueue_tail(pmx_ctx* c, int t) {
    const pmx_adaprox_params& p = c->ada;
    FinishArgs f{};
    f.s = sub_args(c, t);
    f.Xp[0] = c->Xp[0]; f.Xp[1] = s_view(c, 1, c->Xp[1]);
    f.colpart = c->colpart;
    f.check_convergence = p.check_convergence;
    static_assert(EW_BLOCKS == 256, "k_grad_f16_v8 folds 256 partial maxima per factor");
    f.absmax_out = c->f16_scales ? c->absmax : nullptr;
    launch_ada_finish(f, c->stream);
    AdaDecideArgs d{};
    d.al = alpha_args(c);
    d.al.use_fixed = p.use_fixed_steps;
    d.al.fixed[0] = (float)p.fixed_alpha[0];
    d.al.fixed[1] = (float)p.fixed_alpha[1];
    d.partials = c->partials;
    d.e_rel[0] = p.e_rel[0]; d.e_rel[1] = p.e_rel[1];
    // row-sharded: A's sums are only local here; the test is made after the next all-reduce (k_shard_post)
    d.check_convergence = c->comm ? 0 : p.check_convergence;
    d.has_prox[0] = p.prox[0].n > 0; d.has_prox[1] = p.prox[1].n > 0;
    d.host_tau[0] = c->host_tau[0]; d.host_tau[1] = c->host_tau[1];
    launch_ada_decide(d, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

static int ada_enqueue_moment(pmx_ctx* c, int it, double b1t, double b1prev) {
    const pmx_adaprox_params& p = c->ada;
    MomentArgs m{};
    for (int j = 0; j < 2; ++j) {
        m.X[j] = s_view(c, j, c->X[j]); m.Xp[j] = s_view(c, j, c->Xp[j]);
        m.Mm[j] = s_view(c, j, c->Mm[j]); m.Vv[j] = s_view(c, j, c->Vv[j]);
        m.Vh[j] = p.warm_vhat ? s_view(c, j, c->Vh[j]) : nullptr;
        m.Psi[j] = s_view(c, j, c->Psi[j]);
        m.slab[j] = slab_ref(c, j);
        m.rows[j] = upd_rows(c, j);
        m.has_prox[j] = p.prox[j].n > 0 || p.host_prox[j];   // (Psi is kept for a host-side proximal loop as well)
    }
    if (c->shard_grad_from_comm) {   // row-sharded: gSt is the all-reduced sum sitting in the comm buffer (S-split: this rank's chunk of it)
        m.slab[1].base = c->ssplit ? c->comm_out : c->comm;
        m.slab[1].n = 1; m.slab[1].ld = (int)c->K; m.slab[1].extra = nullptr;
    }
    m.K = (int)c->K;
    m.status = c->dstatus;
    m.partials = c->partials;
    m.scheme = p.scheme;
    m.it = it;
    m.b1t = b1t; m.b1prev = b1prev; m.b2 = p.b2; m.eps = p.eps; m.p = p.p;
    m.check_convergence = p.check_convergence;
    launch_ada_moment(m, c->stream);                                      // algorithms.py:375-378
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// the fused iteration tail (k_ada_tail): moment + update, proximal sub-iterations, finish, next step sizes
static int ada_enqueue_tail_fused(pmx_ctx* c, int it, double b1t, double b1prev) {
    const pmx_adaprox_params& p = c->ada;
    TailArgs t{};
    MomentArgs& m = t.m;
    for (int j = 0; j < 2; ++j) {
        m.X[j] = s_view(c, j, c->X[j]); m.Xp[j] = s_view(c, j, c->Xp[j]);
        m.Mm[j] = s_view(c, j, c->Mm[j]); m.Vv[j] = s_view(c, j, c->Vv[j]);
        m.Vh[j] = p.warm_vhat ? s_view(c, j, c->Vh[j]) : nullptr;
        m.Psi[j] = nullptr;
        m.slab[j] = slab_ref(c, j);
        m.rows[j] = upd_rows(c, j);
        m.has_prox[j] = p.prox[j].n > 0;
        t.prox[j] = to_dev(p.prox[j]);
        t.e_rel[j] = p.e_rel[j];
        t.slots[j] = (int)((upd_rows(c, j) + 8191) / 8192);
    }
    if (c->shard_grad_from_comm) { m.slab[1].base = c->ssplit ? c->comm_out : c->comm; m.slab[1].n = 1; m.slab[1].ld = (int)c->K; m.slab[1].extra = nullptr; }
    m.K = (int)c->K;
    m.status = c->dstatus;
    m.partials = c->partials;
    m.scheme = p.scheme;
    m.it = it;
    m.b1t = b1t; m.b1prev = b1prev; m.b2 = p.b2; m.eps = p.eps; m.p = p.p;
    m.check_convergence = p.check_convergence;
    t.prox_max_iter = p.prox_max_iter;
    t.colpart = c->colpart;
    t.absmax_out = c->f16_scales ? c->absmax : nullptr;
    t.al = alpha_args(c);
    t.al.use_fixed = p.use_fixed_steps;
    t.al.fixed[0] = (float)p.fixed_alpha[0];
    t.al.fixed[1] = (float)p.fixed_alpha[1];
    t.decide_check = c->comm ? 0 : 1;     // row-sharded: the outer test is made after the next all-reduce (k_shard_post)
    t.bar = c->gridbar;
    t.prof = c->tailprof;
    t.prof_fine = getenv("PMX_TAIL_PROF") && atoi(getenv("PMX_TAIL_PROF")) == 2;
    // PMX_TAIL_LOCKFILE (tests only): several processes share ONE GPU -- their persistent tails (one workgroup per CU each,
    // a census barrier at the top) cannot be resident together, so each is run to completion under an inter-process lock.
    // The product configuration is one process per GPU and never sets it.
    const char* lockfile = c->hook_tail_lockfile.empty() ? nullptr : c->hook_tail_lockfile.c_str();   // (read once, at pmx_ctx_create)
    int lockfd = -1;
    if (lockfile) {
        lockfd = open(lockfile, O_CREAT | O_RDWR, 0600);
        if (lockfd < 0) FAIL(PMX_E_STATE, "PMX_TAIL_LOCKFILE: cannot open %s", lockfile);
        HIP_CHECK(hipStreamSynchronize(c->stream));          // everything this rank enqueued before the tail has left the GPU
        if (flock(lockfd, LOCK_EX) != 0) { close(lockfd); FAIL(PMX_E_STATE, "PMX_TAIL_LOCKFILE: flock failed"); }
    }
    hipError_t le = launch_ada_tail(t, c->stream);
    if (lockfd >= 0) {
        hipError_t se = hipStreamSynchronize(c->stream);
        flock(lockfd, LOCK_UN);
        close(lockfd);
        HIP_CHECK(se);
    }
    HIP_CHECK(le);
    return PMX_OK;
}

// after_tail: the previous kernels on the stream were THIS call's finish + decide of the preceding iteration
static int ada_enqueue_head(pmx_ctx* c, int it, double b1t, double b1prev, bool after_tail) {
    int rc = enqueue_grad(c, c->X[0], c->X[1], 1, 1, after_tail);         // algorithms.py:369
    if (rc != PMX_OK) return rc;
    return ada_enqueue_moment(c, it, b1t, b1prev);
}

extern "C" int pmx_adaprox_run(pmx_ctx* c, int n_iter, const double* b1, double b1_prev, pmx_result* res) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_ADAPROX) FAIL(PMX_E_STATE, "pmx_adaprox_begin has not been called");
    if (n_iter < 0 || (n_iter > 0 && !b1)) FAIL(PMX_E_INVALID, "bad n_iter / b1");
    for (int i = 0; i < n_iter; ++i)
        if (!(b1[i] >= 0 && b1[i] < 1)) FAIL(PMX_E_INVALID, "b1 out of [0,1)");       // algorithms.py:330
    const pmx_adaprox_params& p = c->ada;
    const bool any_prox = p.prox[0].n > 0 || p.prox[1].n > 0;
    const int it0 = c->hstatus->it_done;
    if (c->f64big) {         // [r6] k_big_f64.hip: two MFMA passes, then the tail as a chain of launches; the proximal loops' lengths are guessed
        auto args_of = [&](int gi) {
            Ada64bArgs b{};
            Ada64Args& a = b.a;
            for (int j = 0; j < 2; ++j) {
                a.X[j] = c->Xd[j]; a.Xp[j] = c->Xpd[j]; a.Mm[j] = c->Md[j]; a.Vv[j] = c->Vd[j];
                a.Vh[j] = p.warm_vhat ? c->Vhd[j] : nullptr;
                a.Psi[j] = c->Psid[j]; a.z[j] = c->zd[j];
                a.slab[j] = c->slabd[j];
                a.nslab[j] = j == 0 ? c->nSlabA : c->nSlabS;
                a.rows[j] = c->rows[j];
                a.prox[j] = to_dev(p.prox[j]);
                a.has_prox[j] = p.prox[j].n > 0;
                a.e_rel[j] = p.e_rel[j];
                a.fixed[j] = p.fixed_alpha[j];
            }
            a.K = (int)c->K;
            a.status = c->dstatus;
            a.scheme = p.scheme;
            a.it = it0 + gi;
            a.b1t = b1[gi]; a.b1prev = gi == 0 ? b1_prev : b1[gi - 1];
            a.b2 = p.b2; a.eps = p.eps; a.p = p.p;
            a.check_convergence = p.check_convergence;
            a.prox_max_iter = p.prox_max_iter;
            a.use_fixed = p.use_fixed_steps;
            a.alpha_out = c->alpha64;
            b.partials = c->partials;
            b.colpart = c->colpart64;
            return b;
        };
        int gi = 0;
        while (gi < n_iter && !c->hstatus->stopped) {
            const int chunk = std::min(n_iter - gi, 16);
            const int nsub = any_prox ? std::max(1, std::min(c->nsub64, p.prox_max_iter)) : 0;
            for (int i = 0; i < chunk; ++i) {
                rc = enqueue_front64(c, c->Xd[0], c->Xd[1], 1, 1, true, false, 1.0);       // algorithms.py:369
                if (rc != PMX_OK) return rc;
                Ada64bArgs b = args_of(gi + i);
                launch_ada64b_head(b, c->stream);                                          // :370-378
                for (int t = 1; t <= nsub; ++t) { b.t = t; launch_ada64b_sub(b, c->stream); }   // :386-392
                b.t = nsub;
                launch_ada64b_close(b, c->stream);                                         // :400-410
                HIP_CHECK(hipGetLastError());
            }
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            if (c->hstatus->halt && c->hstatus->reason == HALT_ERROR) FAIL(PMX_E_HIP, "%s", chain_error_text(c));
            int done = c->hstatus->it_done - it0;                // iterations of this call completed
            int passes = nsub;
            while (c->hstatus->halt && c->hstatus->reason == HALT_NEED_SUB) {      // iteration `done` stands in front of its verdict: more passes
                const int more = std::min(p.prox_max_iter - passes, std::max(passes, 4));
                if (more <= 0) FAIL(PMX_E_STATE, "adaprox (fp64): the proximal loop asks for passes beyond prox_max_iter");
                HIP_CHECK(hipMemsetAsync(&c->dstatus->halt, 0, 2 * sizeof(int), c->stream));      // halt, reason
                Ada64bArgs b = args_of(done);
                for (int t = passes + 1; t <= passes + more; ++t) { b.t = t; launch_ada64b_sub(b, c->stream); }
                passes += more;
                b.t = passes;
                launch_ada64b_close(b, c->stream);
                HIP_CHECK(hipGetLastError());
                rc = read_status(c);
                if (rc != PMX_OK) return rc;
                if (c->hstatus->halt && c->hstatus->reason == HALT_ERROR) FAIL(PMX_E_HIP, "%s", chain_error_text(c));
                done = c->hstatus->it_done - it0;
            }
            if (any_prox) c->nsub64 = std::max(c->hstatus->last_tau[0], c->hstatus->last_tau[1]) + 2;
            gi = done;
        }
        fill_result(c, res, it0);
        return PMX_OK;
    }
    if (c->f64) {            // K1 tiles (k64_front without its step-rule workgroups) + the whole tail by one workgroup (k64_ada_iter)
        for (int left = n_iter, gi = 0; left > 0 && !c->hstatus->stopped; left -= 16) {
            for (int i = 0; i < std::min(left, 16); ++i, ++gi) {
                rc = enqueue_front64(c, c->Xd[0], c->Xd[1], 1, 1, true, false, 1.0);       // algorithms.py:369
                if (rc != PMX_OK) return rc;
                Ada64Args a{};
                for (int j = 0; j < 2; ++j) {
                    a.X[j] = c->Xd[j]; a.Xp[j] = c->Xpd[j]; a.Mm[j] = c->Md[j]; a.Vv[j] = c->Vd[j];
                    a.Vh[j] = p.warm_vhat ? c->Vhd[j] : nullptr;
                    a.Psi[j] = c->Psid[j]; a.z[j] = c->zd[j];
                    a.slab[j] = c->slabd[j];
                    a.nslab[j] = j == 0 ? c->nSlabA : c->nSlabS;
                    a.rows[j] = c->rows[j];
                    a.prox[j] = to_dev(p.prox[j]);
                    a.has_prox[j] = p.prox[j].n > 0;
                    a.e_rel[j] = p.e_rel[j];
                    a.fixed[j] = p.fixed_alpha[j];
                }
                a.K = (int)c->K;
                a.status = c->dstatus;
                a.scheme = p.scheme;
                a.it = it0 + gi;
                a.b1t = b1[gi]; a.b1prev = gi == 0 ? b1_prev : b1[gi - 1];
                a.b2 = p.b2; a.eps = p.eps; a.p = p.p;
                a.check_convergence = p.check_convergence;
                a.prox_max_iter = p.prox_max_iter;
                a.use_fixed = p.use_fixed_steps;
                a.alpha_out = c->alpha64;
                launch_ada64_iter(a, c->stream);                                           // algorithms.py:370-410
                HIP_CHECK(hipGetLastError());
            }
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            if (c->hstatus->halt && c->hstatus->reason == HALT_ERROR) FAIL(PMX_E_HIP, "%s", chain_error_text(c));
        }
        fill_result(c, res, it0);
        return PMX_OK;
    }
    int done = 0;            // iterations of this call completed
    int tails = 0;           // iteration tails enqueued by this call (their finish kernel leaves the factor maxima behind)
    while (done < n_iter && !c->hstatus->stopped) {
        // ---- enqueue a chunk of whole iterations ------------------------------------------------
        // (the fused tail decides its proximal loops on the device: nothing to speculate on, longer chunks between host syncs)
        const int chunk = std::min(n_iter - done, c->tail_fused ? 64 : 16);
        const int nsub = any_prox ? std::max(1, std::min(c->nsub_guess, p.prox_max_iter)) : 0;
        // passes per launch for this chunk: 4 when the loops have been ending within 4 passes (the usual steady state:
        // 1 pass for a projection, 2-3 for prox_unity_plus), else 8; PMX_SUB_BATCH=1 keeps one pass per launch
        if (c->sub_nt != 1) c->sub_nt = nsub <= 4 ? 4 : SUB_NT_MAX;
        for (int i = 0; i < chunk; ++i) {
            const int gi = done + i;
            if (c->tail_fused) {
                rc = enqueue_grad(c, c->X[0], c->X[1], 1, 1, tails > 0);
                if (rc == PMX_OK) rc = ada_enqueue_tail_fused(c, it0 + gi, b1[gi], gi == 0 ? b1_prev : b1[gi - 1]);
                ++tails;
                if (rc != PMX_OK) return rc;
                continue;
            }
            rc = ada_enqueue_head(c, it0 + gi, b1[gi], gi == 0 ? b1_prev : b1[gi - 1], tails > 0);
            if (rc != PMX_OK) return rc;
            rc = ada_enqueue_tail(c, ada_enqueue_subs(c, 0, nsub));
            ++tails;
            if (rc != PMX_OK) return rc;
        }
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        // ---- a chain stopped inside an iteration's sub-iteration loop: feed it more passes --------
        const int nsub_enq = nsub > 0 ? ((nsub + c->sub_nt - 1) / c->sub_nt) * c->sub_nt : 0;
        int t_enq = nsub_enq;
        while (c->hstatus->halt && c->hstatus->reason == HALT_NEED_SUB) {
            rc = clear_halt(c);
            if (rc != PMX_OK) return rc;
            const int more = std::min(std::max(4, t_enq), 64);
            t_enq = ada_enqueue_subs(c, t_enq, more);
            rc = ada_enqueue_tail(c, t_enq);
            if (rc != PMX_OK) return rc;
            // the iterations that followed in the chunk were skipped: re-enqueue them after this one
            const int finished_if_ok = c->hstatus->it_done - it0 + 1;
            for (int gi = finished_if_ok; gi < done + chunk; ++gi) {
                rc = ada_enqueue_head(c, it0 + gi, b1[gi], gi == 0 ? b1_prev : b1[gi - 1], true);
                if (rc != PMX_OK) return rc;
                rc = ada_enqueue_tail(c, ada_enqueue_subs(c, 0, nsub));
                if (rc != PMX_OK) return rc;
            }
            const int it_before = c->hstatus->it_done;
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            if (c->hstatus->it_done > it_before) t_enq = nsub_enq;   // moved on to a later iteration
        }
        done = c->hstatus->it_done - it0;
        {
            int again = 0;
            rc = chain_fault_fallback(c, &again);
            if (rc != PMX_OK) return rc;
            if (again) { tails = 0; continue; }
        }
        if (any_prox) c->nsub_guess = std::max(2, std::max(c->hstatus->last_tau[0], c->hstatus->last_tau[1]));
        if (c->hstatus->halt && c->hstatus->reason == HALT_ERROR) FAIL(PMX_E_HIP, "%s", chain_error_text(c));
    }
    if (c->tailprof && c->tail_fused) {
        long long h[16];
        HIP_CHECK(hipMemcpy(h, c->tailprof, sizeof(h), hipMemcpyDeviceToHost));
        static const char* nm[] = {"census", "moment", "B1", "maxpsi", "sub", "B2", "judge+replay", "finish", "B3", "decide"};
        static const char* nf[] = {"census", "moment", "B1", "maxpsi", "passesA", "sumsA", "passesS", "sumsS", "(end)", "B2", "judge+replay", "finish", "B3", "decide"};
        const bool fine = getenv("PMX_TAIL_PROF") && atoi(getenv("PMX_TAIL_PROF")) == 2;
        const int np = fine ? 14 : 10;
        fprintf(stderr, "[tailprof] us:");
        for (int i = 0; i < np; ++i) fprintf(stderr, " %s=%.2f", fine ? nf[i] : nm[i], (double)(h[i + 1] - h[i]) / 100.0);
        fprintf(stderr, " total=%.2f\n", (double)(h[np] - h[0]) / 100.0);
    }
    fill_result(c, res, it0);
    return PMX_OK;
}

extern "C" int pmx_adaprox_set_alpha(pmx_ctx* c, const float* alpha) {
    REJECT_F64(c);
    if (!c || !alpha) FAIL(PMX_E_INVALID, "NULL argument");
    if (c->algo != ALG_ADAPROX || c->ada.use_fixed_steps != 2) FAIL(PMX_E_STATE, "pmx_adaprox_begin with use_fixed_steps = 2 has not been called");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipMemcpyAsync(&c->dstatus->alpha[0][0], alpha, sizeof(float) * c->K, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(&c->dstatus->alpha[1][0], alpha + c->K, sizeof(float) * c->K, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_adaprox_split(pmx_ctx* c, int phase, int it, double b1_it, double b1_prev, const int* host_tau, double* maxpsi, pmx_result* res) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_ADAPROX) FAIL(PMX_E_STATE, "pmx_adaprox_begin has not been called");
    if (c->tail_fused) FAIL(PMX_E_STATE, "pmx_adaprox_split needs a context begun with host_prox");
    if (!(b1_it >= 0 && b1_it < 1)) FAIL(PMX_E_INVALID, "b1 out of [0,1)");
    const pmx_adaprox_params& p = c->ada;
    if (phase == 0) {
        { rc = one_iteration_per_call(c); if (rc != PMX_OK) return rc; }
        rc = ada_enqueue_head(c, it, b1_it, b1_prev, false);
        if (rc != PMX_OK) return rc;
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        if (maxpsi) {                     // np.max(Psi) per block (NaN lets through): fold of the per-workgroup partials
            std::vector<double> h(2 * EW_BLOCKS);
            HIP_CHECK(hipMemcpy(h.data(), c->partials + (size_t)SL_MAXPSI * 2 * EW_BLOCKS, sizeof(double) * 2 * EW_BLOCKS, hipMemcpyDeviceToHost));
            for (int j = 0; j < 2; ++j) {
                double m = -1.0;
                for (int b = 0; b < EW_BLOCKS; ++b) { const double v = h[j * EW_BLOCKS + b]; m = (m != m || v != v) ? NAN : std::max(m, v); }
                maxpsi[j] = m;
            }
        }
        return PMX_OK;
    }
    if (phase != 1) FAIL(PMX_E_INVALID, "bad phase %d", phase);
    c->host_tau[0] = host_tau ? host_tau[0] : 0;
    c->host_tau[1] = host_tau ? host_tau[1] : 0;
    const bool any_prox = p.prox[0].n > 0 || p.prox[1].n > 0;
    const int it0 = c->hstatus->it_done;
    const int nsub = any_prox ? std::max(1, std::min(c->nsub_guess, p.prox_max_iter)) : 0;
    if (c->sub_nt != 1) c->sub_nt = nsub <= 4 ? 4 : SUB_NT_MAX;
    int t_enq = ada_enqueue_subs(c, 0, nsub);
    rc = ada_enqueue_tail(c, t_enq);
    if (rc != PMX_OK) return rc;
    rc = read_status(c);
    if (rc != PMX_OK) return rc;
    while (c->hstatus->halt && c->hstatus->reason == HALT_NEED_SUB) {
        rc = clear_halt(c);
        if (rc != PMX_OK) return rc;
        t_enq = ada_enqueue_subs(c, t_enq, std::min(std::max(4, t_enq), 64));
        rc = ada_enqueue_tail(c, t_enq);
        if (rc != PMX_OK) return rc;
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
    }
    if (any_prox) c->nsub_guess = std::max(2, std::max(c->hstatus->last_tau[0], c->hstatus->last_tau[1]));
    c->host_tau[0] = c->host_tau[1] = 0;
    if (c->hstatus->halt && c->hstatus->reason == HALT_ERROR) FAIL(PMX_E_HIP, "%s", chain_error_text(c));
    fill_result(c, res, it0);
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
// block-SDMM                                              (proxmin/algorithms.py:653-850)
// ------------------------------------------------------------------------------------------------
extern "C" int pmx_bsdmm_begin(pmx_ctx* c, const pmx_bsdmm_params* p) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (!p) FAIL(PMX_E_INVALID, "params is NULL");
    if (c->W || c->Wd_on)                              // bsdmm's steps come from nmf.step_pgm (nmf.py:187-193)
        FAIL(PMX_E_INVALID, "The truth value of an array with more than one element is ambiguous. Use a.any() or a.all()");
    for (int j = 0; j < 2; ++j) {
        rc = check_prox(p->prox_f[j], j ? "prox_S" : "prox_A");
        if (rc != PMX_OK) return rc;
        if (p->n_g[j] < 0 || p->n_g[j] > PMX_MAX_G) FAIL(PMX_E_UNSUPPORTED, "at most %d constraints per block", PMX_MAX_G);
        for (int i = 0; i < p->n_g[j]; ++i) {
            rc = check_prox(p->prox_g[j][i], "proxs_g");
            if (rc != PMX_OK) return rc;
        }
    }
    if (p->n_order < 0 || p->n_order > 8) FAIL(PMX_E_INVALID, "update_order: at most 8 entries");
    for (int i = 0; i < p->n_order; ++i)
        if (p->order[i] != 0 && p->order[i] != 1) FAIL(PMX_E_INVALID, "update_order: block %d out of range", p->order[i]);
    c->bsd = *p;
    c->algo = ALG_BSDMM;
    c->it = 0;
    rc = reset_status(c);
    if (rc != PMX_OK) return rc;
    // utils.initZU (utils.py:244-254): Z_i = copy of X, U_i = 0
    if (c->f64) {
        for (int j = 0; j < 2; ++j) {
            const size_t n = (size_t)c->rows[j] * c->K;
            for (int i = 0; i < p->n_g[j]; ++i) {
                rc = dallocT(c, &c->Zd[j][i], n, false);
                if (rc == PMX_OK) rc = dallocT(c, &c->Ud[j][i], n, false);
                if (rc != PMX_OK) return rc;
                HIP_CHECK(hipMemcpyAsync(c->Zd[j][i], c->Xd[j], n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                HIP_CHECK(hipMemsetAsync(c->Ud[j][i], 0, n * sizeof(double), c->stream));
            }
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        return PMX_OK;
    }
    for (int j = 0; j < 2; ++j) {
        const size_t n = (size_t)c->rows[j] * c->K;
        for (int i = 0; i < p->n_g[j]; ++i) {
            rc = dallocT(c, &c->Zg[j][i], n, false);
            if (rc == PMX_OK) rc = dallocT(c, &c->Ug[j][i], n, false);
            if (rc != PMX_OK) return rc;
            HIP_CHECK(hipMemcpyAsync(c->Zg[j][i], c->X[j], n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIP_CHECK(hipMemsetAsync(c->Ug[j][i], 0, n * sizeof(float), c->stream));
        }
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

static int bsdmm_enqueue_iteration(pmx_ctx* c) {
    const pmx_bsdmm_params& p = c->bsd;
    static const int default_order[2] = {0, 1};
    const int n_order = p.n_order > 0 ? p.n_order : 2;
    const int* order = p.n_order > 0 ? p.order : default_order;
    for (int o = 0; o < n_order; ++o) {                                    // Gauss-Seidel in update_order, algorithms.py:805
        const int j = order[o];
        if (c->f64) {       // K1 tiles of block j + both step rules in one launch (k64_front), the block update by one workgroup
            int rc = enqueue_front64(c, c->Xd[0], c->Xd[1], j == 0, j == 1, true, true, 1.0);   // nmf.py:181-193
            if (rc != PMX_OK) return rc;
            Bsdmm64Args u{};
            u.X = c->Xd[j];
            u.slab = c->slabd[j];
            u.nslab = j == 0 ? c->nSlabA : c->nSlabS;
            for (int i = 0; i < p.n_g[j]; ++i) { u.Z[i] = c->Zd[j][i]; u.U[i] = c->Ud[j][i]; u.prox_g[i] = to_dev(p.prox_g[j][i]); }
            u.rows = c->rows[j];
            u.K = (int)c->K;
            u.j = j;
            u.n_g = p.n_g[j];
            u.prox_f = to_dev(p.prox_f[j]);
            u.status = c->dstatus;
            u.e_rel = p.e_rel[j];
            u.e_abs = p.e_abs[j];
            u.last_block = o == n_order - 1;
            if (c->f64big) {             // [r6] the update over the grid, its sums folded by the fp32 path's decide kernel (it only ever saw fp64 sums)
                launch_bsdmm64b_update(u, c->partials, c->stream);
                BsdmmDecideArgs d{};
                d.status = c->dstatus; d.partials = c->partials;
                d.j = j; d.n_g = p.n_g[j];
                d.size = c->rows[j] * c->K;
                d.e_rel = p.e_rel[j]; d.e_abs = p.e_abs[j];
                d.last_block = u.last_block;
                launch_bsdmm_decide(d, c->stream);
            } else launch_bsdmm64_block(u, c->stream);
            HIP_CHECK(hipGetLastError());
            continue;
        }
        // (the step of block j comes from the OTHER factor's Gram matrix; its partials are there if that factor's update left them)
        int rc = enqueue_steps(c, c->X[0], c->X[1], j == 0, j == 1, 1.0, c->gram_fresh[1 - j]);   // nmf.py:187-193
        if (rc != PMX_OK) return rc;
        rc = enqueue_grad(c, c->X[0], c->X[1], j == 0, j == 1, c->absmax_by_finish);   // nmf.py:181-185 (only grads[j] is used)
        if (rc != PMX_OK) return rc;
        BsdmmArgs u{};
        // [r4] the partial Gram matrices of the new X_j from this launch (BsdmmArgs::gramPart): K <= 64, factors of <= 16384 rows
        // (cfg5); PMX_GRAM_IN_UPDATE=0 keeps k_gram_partial (A/B)
        const bool gram_here = c->gram_in_update && !eig_small_applies(c) && c->K <= 64 && gram_per(c->rows[j]) <= BSDMM_GRAM_ROWS;
        u.gramPart = gram_here ? c->gramPart : nullptr;
        u.KP = c->KP;
        c->gram_fresh[j] = gram_here;
        u.X = c->X[j];
        u.slab = slab_ref(c, j);
        for (int i = 0; i < p.n_g[j]; ++i) { u.Z[i] = c->Zg[j][i]; u.U[i] = c->Ug[j][i]; u.prox_g[i] = to_dev(p.prox_g[j][i]); }
        u.rows = c->rows[j];
        u.K = (int)c->K;
        u.j = j;
        u.n_g = p.n_g[j];
        u.prox_f = to_dev(p.prox_f[j]);
        u.status = c->dstatus;
        u.partials = c->partials;
        // this block's maxima for the fp16 K1 (the other block's are still those of the launch that last wrote it)
        u.absmax_out = c->f16_scales ? c->absmax : nullptr;
        launch_bsdmm_update(u, c->stream);
        c->absmax_by_finish = u.absmax_out != nullptr;
        BsdmmDecideArgs d{};
        d.status = c->dstatus;
        d.partials = c->partials;
        d.j = j;
        d.n_g = p.n_g[j];
        d.size = c->rows[j] * c->K;
        d.e_rel = p.e_rel[j];
        d.e_abs = p.e_abs[j];
        d.last_block = o == n_order - 1;
        // [r4] the test rides in the next step rule's k_gram_reduce launch (one launch less per block); pmx_bsdmm_run flushes the
        // last one of a chunk before it reads the status
        c->bsd_decide = d;
        c->bsd_decide_pending = true;
        HIP_CHECK(hipGetLastError());
    }
    return PMX_OK;
}

// One block update of ONE bsdmm iteration in pieces, for user-defined operators (host round trips; see BsdmmArgs::stage):
//   phase 0  step_f of block j (res->steps[j]) and its gradient; with host_f the argument of prox_f -> PMX_BUF_TMP_*
//   phase 1  X_j <- prox_f(..) (the device operator, or PMX_BUF_TMP_* as the caller left it); without user members in
//            proxs_g[j] also the constraint updates, else their arguments X_j + U_i -> PMX_BUF_TG0 + ..
//   phase 2  the constraint updates with the user members' results taken from PMX_BUF_TG0 + .. (if any), the Boyd test of
//            the block; last_block: end-of-iteration bookkeeping (iteration counter, stop when every block has converged)
extern "C" int pmx_bsdmm_split(pmx_ctx* c, int j, int phase, int host_f, unsigned host_g, int last_block, double step_f_host, pmx_result* res) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_BSDMM) FAIL(PMX_E_STATE, "pmx_bsdmm_begin has not been called");
    if (j != 0 && j != 1) FAIL(PMX_E_INVALID, "block %d out of range", j);
    const pmx_bsdmm_params& p = c->bsd;
    if (host_g >> p.n_g[j]) FAIL(PMX_E_INVALID, "host_g names a constraint that does not exist");
    // step_f_host: 0 = "no user step" (the device's Lipschitz rule); anything else must be a usable step -- a negative or NaN
    // value from a user steps_f_cb must not fall through to the NMF rule of a context that may have no NMF gradient at all
    if (phase == 0 && !(step_f_host >= 0.0)) FAIL(PMX_E_INVALID, "block %d: the user step %g is not a positive number", j, step_f_host);
    auto args = [&](int stage) {
        BsdmmArgs u{};
        u.X = c->X[j];
        u.slab = slab_ref(c, j);
        for (int i = 0; i < p.n_g[j]; ++i) { u.Z[i] = c->Zg[j][i]; u.U[i] = c->Ug[j][i]; u.prox_g[i] = to_dev(p.prox_g[j][i]); u.T[i] = c->Tg[j][i]; }
        u.rows = c->rows[j];
        u.K = (int)c->K;
        u.j = j;
        u.n_g = p.n_g[j];
        u.prox_f = to_dev(p.prox_f[j]);
        u.status = c->dstatus;
        u.partials = c->partials;
        u.absmax_out = nullptr;
        u.stage = stage;
        u.host_f = host_f;
        u.host_g = host_g;
        u.Tf = c->Xp[j];
        return u;
    };
    const int it0 = c->hstatus->it_done;
    switch (phase) {
        case 0: {
            { rc = one_iteration_per_call(c); if (rc != PMX_OK) return rc; }
            if (host_f) { rc = dallocT(c, &c->Xp[j], (size_t)c->rows[j] * c->K, false); if (rc != PMX_OK) return rc; }
            for (int i = 0; i < p.n_g[j]; ++i)
                if ((host_g >> i) & 1u) { rc = dallocT(c, &c->Tg[j][i], (size_t)c->rows[j] * c->K, false); if (rc != PMX_OK) return rc; }
            if (step_f_host > 0.0) {       // a user steps_f_cb (algorithms.py:807): its value for this block
                HIP_CHECK(hipMemcpyAsync(&c->dstatus->step[j], &step_f_host, sizeof(double), hipMemcpyHostToDevice, c->stream));
                HIP_CHECK(hipStreamSynchronize(c->stream));      // (the source is this call's argument)
            } else {
                rc = enqueue_steps(c, c->X[0], c->X[1], j == 0, j == 1, 1.0);   // nmf.py:187-193
            }
            if (rc == PMX_OK) rc = enqueue_grad(c, c->X[0], c->X[1], j == 0, j == 1);
            if (rc != PMX_OK) return rc;
            if (host_f) launch_bsdmm_update(args(1), c->stream);
            HIP_CHECK(hipGetLastError());
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            if (res) { res->steps[0] = c->hstatus->step[0]; res->steps[1] = c->hstatus->step[1]; }
            return PMX_OK;
        }
        case 1:
            launch_bsdmm_update(args(2), c->stream);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(c->stream));
            return PMX_OK;
        case 2: {
            if (host_g) launch_bsdmm_update(args(3), c->stream);
            BsdmmDecideArgs d{};
            d.status = c->dstatus;
            d.partials = c->partials;
            d.j = j;
            d.n_g = p.n_g[j];
            d.size = c->rows[j] * c->K;
            d.e_rel = p.e_rel[j];
            d.e_abs = p.e_abs[j];
            d.last_block = last_block;
            launch_bsdmm_decide(d, c->stream);
            HIP_CHECK(hipGetLastError());
            rc = read_status(c);
            if (rc != PMX_OK) return rc;
            fill_result(c, res, it0);
            return PMX_OK;
        }
        default: FAIL(PMX_E_INVALID, "bad phase %d", phase);
    }
}

extern "C" int pmx_bsdmm_run(pmx_ctx* c, int n_iter, pmx_result* res) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c, true);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_BSDMM) FAIL(PMX_E_STATE, "pmx_bsdmm_begin has not been called");
    if (n_iter < 0) FAIL(PMX_E_INVALID, "n_iter < 0");
    const int it0 = c->hstatus->it_done;
    c->gram_fresh[0] = c->gram_fresh[1] = false;     // (the caller may have touched the factors since the last call)
    c->bsd_decide_pending = false;
    int left = n_iter;
    while (left > 0 && !c->hstatus->stopped) {
        const int chunk = std::min(left, 16);
        for (int i = 0; i < chunk; ++i) {
            rc = bsdmm_enqueue_iteration(c);
            if (rc != PMX_OK) return rc;
        }
        rc = bsdmm_flush_decide(c);
        if (rc != PMX_OK) return rc;
        rc = read_status(c);
        if (rc != PMX_OK) return rc;
        int again = 0;
        rc = c->f64 ? PMX_OK : chain_fault_fallback(c, &again);
        if (rc != PMX_OK) return rc;
        if (again) {
            left = n_iter - (c->hstatus->it_done - it0);
            c->gram_fresh[0] = c->gram_fresh[1] = false;
            continue;
        }
        left -= chunk;
    }
    fill_result(c, res, it0);
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------------
// row-sharded multi-GPU; see include/pmx.h for the protocol
// ------------------------------------------------------------------------------------------------
static int64_t comm_count(const pmx_ctx* c) { return c->N * c->K + (int64_t)c->KP * c->KP + MAXK + 32; }

extern "C" int pmx_set_world(pmx_ctx* c, int rank, int world, int64_t M_global) {
    REJECT_F64(c);
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (world < 1 || rank < 0 || rank >= world) FAIL(PMX_E_INVALID, "bad rank/world");
    if (M_global < c->M) FAIL(PMX_E_INVALID, "M_global < local M");
    c->rank = rank; c->world = world; c->M_global = M_global;
    return PMX_OK;
}

extern "C" int pmx_set_host_grad(pmx_ctx* c, int on) {
    REJECT_F64(c);
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    HIP_CHECK(hipSetDevice(c->device));
    if (on) {
        if (c->W) FAIL(PMX_E_UNSUPPORTED, "a host-side gradient and device-side weights do not go together");
        { int rc = one_iteration_per_call(c); if (rc != PMX_OK) return rc; }
        c->haveY = true;                   // nothing M x N is needed: the fused residual kernel never runs in this context
    }
    c->host_grad = on != 0;
    return PMX_OK;
}

extern "C" int pmx_comm_layout(pmx_ctx* c, int64_t* count, int64_t offsets[3]) {
    if (!c || !count || !offsets) FAIL(PMX_E_INVALID, "NULL argument");
    *count = comm_count(c);
    offsets[0] = c->N * c->K;                               // Gram(A)
    offsets[1] = offsets[0] + (int64_t)c->KP * c->KP;       // colsum(A)
    offsets[2] = offsets[1] + MAXK;                         // scalars
    return PMX_OK;
}

extern "C" int pmx_set_comm_buffer(pmx_ctx* c, float* dptr, int64_t count) {
    REJECT_F64(c);
    if (!c || !dptr) FAIL(PMX_E_INVALID, "NULL argument");
    if (count < comm_count(c)) FAIL(PMX_E_INVALID, "comm buffer too small: %lld < %lld", (long long)count, (long long)comm_count(c));
    if (c->W) FAIL(PMX_E_UNSUPPORTED, "weights are not supported in row-sharded runs");
    c->comm = dptr;
    c->comm_gram_dirty = true;
    return PMX_OK;
}

static AlphaArgs alpha_args_sharded(pmx_ctx* c) {
    AlphaArgs a = alpha_args(c);
    a.use_fixed = c->ada.use_fixed_steps;
    a.fixed[0] = (float)c->ada.fixed_alpha[0];
    a.fixed[1] = (float)c->ada.fixed_alpha[1];
    a.comm_colsum = c->comm + c->N * c->K + (int64_t)c->KP * c->KP;
    return a;
}

static int shard_pack(pmx_ctx* c, int fold_grad, bool with_gram = false, int n_extra = 0) {
    PackArgs p{};
    p.slabS = slab_ref(c, 1);
    p.comm = c->comm;
    p.N = c->N;
    p.K = (int)c->K; p.KP = c->KP;
    p.colpart = c->colpart;
    p.partials = c->partials;
    p.gramA = with_gram ? c->gramG : nullptr;   // factor 0 = A
    p.status = c->dstatus;
    p.fold_grad = fold_grad;
    p.n_extra = n_extra;
    launch_shard_pack(p, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

static int shard_post(pmx_ctx* c, int have_prev) {
    ShardPostArgs q{};
    const bool ada = c->algo == ALG_ADAPROX;
    q.al = ada ? alpha_args_sharded(c) : alpha_args(c);
    q.scalars = c->comm + c->N * c->K + (int64_t)c->KP * c->KP + MAXK;
    q.partials = c->partials;
    q.e_rel[0] = ada ? c->ada.e_rel[0] : c->pgm.e_rel[0];
    q.e_rel[1] = ada ? c->ada.e_rel[1] : c->pgm.e_rel[1];
    q.check_convergence = ada ? c->ada.check_convergence : 1;
    q.do_alpha = ada;
    q.have_prev = have_prev;
    launch_shard_post(q, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

// ---- S-split: the update of S sharded as well (VERDICT r2 task 3a) ----------------------------
// An all-reduce is a reduce-scatter followed by an all-gather.  With a projection-type prox_S (its result does not depend on
// the number of proximal passes: no cross-rank sum inside the loop) the S update can sit BETWEEN the two halves: each rank
// receives the summed gS of its N / world columns only, updates those columns and their moments, and the columns are
// all-gathered.  Same bytes on the wire, the replicated tail work divided by the rank count.
//   comm buffer = world chunks of [ gSt rows of rank q (N/world x K) | Gram (KP^2) | colsum(A) | colsum(S) | scalars ]
//   (the small sums are copied into every chunk: each rank finds them, summed, in the chunk the reduce-scatter hands it)
extern "C" int pmx_set_s_split(pmx_ctx* c, int on) {
    REJECT_F64(c);
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (!on) { c->ssplit = false; c->scol0 = 0; c->sncol = 0; return PMX_OK; }
    if (c->world < 1) FAIL(PMX_E_STATE, "pmx_set_world has not been called");
    if (c->N % c->world != 0) FAIL(PMX_E_UNSUPPORTED, "S-split needs N (%lld) divisible by the rank count (%d)", (long long)c->N, c->world);
    c->ssplit = true;
    c->sncol = c->N / c->world;
    c->scol0 = c->sncol * c->rank;
    return PMX_OK;
}
extern "C" int pmx_comm_layout_split(pmx_ctx* c, int64_t* count, int64_t* chunk, int64_t offsets[4]) {
    if (!c || !count || !chunk || !offsets) FAIL(PMX_E_INVALID, "NULL argument");
    if (!c->ssplit) FAIL(PMX_E_STATE, "pmx_set_s_split has not been called");
    *chunk = split_chunk(c);
    *count = *chunk * c->world;
    offsets[0] = c->sncol * c->K;                           // Gram
    offsets[1] = offsets[0] + (int64_t)c->KP * c->KP;       // colsum(A)
    offsets[2] = offsets[1] + MAXK;                         // colsum(S)
    offsets[3] = offsets[2] + MAXK;                         // scalars
    return PMX_OK;
}
extern "C" int pmx_set_comm_out(pmx_ctx* c, float* dptr, int64_t count) {
    if (!c || !dptr) FAIL(PMX_E_INVALID, "NULL argument");
    if (!c->ssplit) FAIL(PMX_E_STATE, "pmx_set_s_split has not been called");
    if (count < split_chunk(c)) FAIL(PMX_E_INVALID, "reduce-scatter output too small: %lld < %lld", (long long)count, (long long)split_chunk(c));
    c->comm_out = dptr;
    return PMX_OK;
}
static int shard_pack_split(pmx_ctx* c, int fold_grad, bool with_gram = false) {
    PackSplitArgs p{};
    p.gramA = with_gram ? c->gramG : nullptr;       // factor 0 = A (pgm: reduced with gS, read back by shard_gram_in)
    p.slabS = slab_ref(c, 1);
    p.comm = c->comm;
    p.N = c->N;
    p.K = (int)c->K; p.KP = c->KP;
    p.colpart = c->colpart;
    p.partials = c->partials;
    p.status = c->dstatus;
    p.fold_grad = fold_grad;
    p.world = c->world;
    p.sncol = c->sncol;
    p.chunk = split_chunk(c);
    p.zero_gram = c->comm_gram_dirty ? 1 : 0;
    c->comm_gram_dirty = false;
    launch_shard_pack_split(p, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
static int shard_post_split(pmx_ctx* c, int have_prev) {
    ShardPostSplitArgs q{};
    q.status = c->dstatus;
    q.extras = c->comm_out + c->sncol * c->K + (int64_t)c->KP * c->KP;
    q.rows_global[0] = c->M_global;
    q.rows_global[1] = c->N;
    q.K = (int)c->K;
    const bool ada = c->algo == ALG_ADAPROX;
    q.use_fixed = ada ? c->ada.use_fixed_steps : 2;              // pgm: no adaprox step sizes to derive (2: leave DevStatus::alpha alone)
    q.fixed[0] = (float)c->ada.fixed_alpha[0];
    q.fixed[1] = (float)c->ada.fixed_alpha[1];
    q.e_rel[0] = ada ? c->ada.e_rel[0] : c->pgm.e_rel[0];
    q.e_rel[1] = ada ? c->ada.e_rel[1] : c->pgm.e_rel[1];
    q.check_convergence = ada ? c->ada.check_convergence : 1;
    q.have_prev = have_prev;
    launch_shard_post_split(q, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_adaprox_phase(pmx_ctx* c, int phase, int it, double b1_it, double b1_prev, int nsub) {
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_ADAPROX) FAIL(PMX_E_STATE, "pmx_adaprox_begin has not been called");
    if (!c->comm) FAIL(PMX_E_STATE, "pmx_set_comm_buffer has not been called");
    const pmx_adaprox_params& p = c->ada;
    for (int i = 0; i < p.prox[0].n; ++i) {
        // A's proximal loop would need cross-rank sums per pass unless the operator is a coordinate-wise
        // projection whose fixed point is reached in one pass (result independent of gamma)
        const pmx_prox& q = p.prox[0].seq[i];
        const bool box = q.op == PMX_PROX_ID || q.op == PMX_PROX_ZERO || q.op == PMX_PROX_PLUS ||
                         ((q.op == PMX_PROX_MIN || q.op == PMX_PROX_MAX || q.op == PMX_PROX_HARD || q.op == PMX_PROX_HARD_PLUS) && !q.relative);
        if (!box) FAIL(PMX_E_UNSUPPORTED, "row-sharded adaprox supports only projection-type prox_A (plus/id/zero/absolute min,max,hard)");
    }
    if (c->ssplit) {
        if (!c->comm_out) FAIL(PMX_E_STATE, "pmx_set_comm_out has not been called");
        for (int i = 0; i < p.prox[1].n; ++i) {      // the same condition for S: its loop must not need a sum over all of S
            const pmx_prox& q = p.prox[1].seq[i];
            const bool box = q.op == PMX_PROX_ID || q.op == PMX_PROX_ZERO || q.op == PMX_PROX_PLUS ||
                             ((q.op == PMX_PROX_MIN || q.op == PMX_PROX_MAX || q.op == PMX_PROX_HARD || q.op == PMX_PROX_HARD_PLUS) && !q.relative);
            if (!box) FAIL(PMX_E_UNSUPPORTED, "S-split supports only projection-type prox_S; use the replicated S update");
        }
    }
    switch (phase) {
        case 0:
            rc = phase_mark(c, 0);
            if (rc == PMX_OK) rc = enqueue_grad(c, c->X[0], c->X[1], 1, 1, c->absmax_by_finish);
            if (rc == PMX_OK) rc = phase_mark(c, 1);
            if (rc != PMX_OK) return rc;
            rc = c->ssplit ? shard_pack_split(c, 1) : shard_pack(c, 1);
            return rc == PMX_OK ? phase_mark(c, 2) : rc;
        case 1: {
            rc = phase_mark(c, 3);
            if (rc == PMX_OK) rc = c->ssplit ? shard_post_split(c, it > 0) : shard_post(c, it > 0);
            if (rc == PMX_OK) rc = phase_mark(c, 4);
            if (rc != PMX_OK) return rc;
            if (c->tail_fused) {
                c->shard_grad_from_comm = true;
                rc = ada_enqueue_tail_fused(c, it, b1_it, b1_prev);
                c->shard_grad_from_comm = false;
                // S-split: the tail saw this rank's columns of S only; the next K1 measures the operand maxima itself
                c->absmax_by_finish = rc == PMX_OK && !c->ssplit;
                return rc == PMX_OK ? phase_mark(c, 5) : rc;
            }
            c->shard_grad_from_comm = true;
            rc = ada_enqueue_moment(c, it, b1_it, b1_prev);
            c->shard_grad_from_comm = false;
            if (rc != PMX_OK) return rc;
            const bool any_prox = p.prox[0].n > 0 || p.prox[1].n > 0;
            const int ns = any_prox ? std::max(1, std::min(nsub, p.prox_max_iter)) : 0;
            if (c->sub_nt != 1) c->sub_nt = ns <= 4 ? 4 : SUB_NT_MAX;   // fixed for this iteration (pmx_adaprox_more_subs continues with it)
            pmx_ctx::SubRec& r = c->sub_rec[it & 63];
            r.it = it; r.nt = c->sub_nt;
            r.enq = ada_enqueue_subs(c, 0, ns);
            rc = ada_enqueue_tail(c, r.enq);
            c->absmax_by_finish = rc == PMX_OK && !c->ssplit;
            return rc == PMX_OK ? phase_mark(c, 5) : rc;
        }
        case 2: return c->ssplit ? shard_pack_split(c, 0) : shard_pack(c, 0);
        case 3: return c->ssplit ? shard_post_split(c, 1) : shard_post(c, 1);
        default: FAIL(PMX_E_INVALID, "bad phase %d", phase);
    }
}

static int pgm_enqueue_update(pmx_ctx* c, bool gS_from_comm, int check) {
    const pmx_pgm_params& p = c->pgm;
    PgmArgs u{};
    for (int j = 0; j < 2; ++j) {            // S-split: block 1 is this rank's sncol columns of S (rows of S^T) only
        u.X[j] = s_view(c, j, c->X[j]);
        u.Xe[j] = s_view(c, j, p.accelerated ? c->Xe[j] : c->X[j]);
        u.G[j] = s_view(c, j, c->G[j]);
        u.slab[j] = slab_ref(c, j);
        u.rows[j] = upd_rows(c, j);
        u.prox[j] = to_dev(p.prox[j]);
    }
    if (gS_from_comm) { u.slab[1].base = c->ssplit ? c->comm_out : c->comm; u.slab[1].n = 1; u.slab[1].ld = (int)c->K; u.slab[1].extra = nullptr; }
    u.K = (int)c->K;
    u.status = c->dstatus;
    u.partials = c->partials;
    u.accelerated = p.accelerated;
    u.omega_next = next_omega(c);
    launch_pgm_update(u, c->stream);
    DecideArgs d{};
    d.status = c->dstatus;
    d.partials = c->partials;
    d.e_rel[0] = p.e_rel[0]; d.e_rel[1] = p.e_rel[1];
    d.check = check;
    launch_pgm_decide(d, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_pgm_phase(pmx_ctx* c, int phase, int it) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_PGM) FAIL(PMX_E_STATE, "pmx_pgm_begin has not been called");
    if (!c->comm) FAIL(PMX_E_STATE, "pmx_set_comm_buffer has not been called");
    const pmx_pgm_params& p = c->pgm;
    if (p.bb_type || p.backtracking) FAIL(PMX_E_UNSUPPORTED, "row-sharded pgm supports the Lipschitz rule or fixed steps only");
    if (c->ssplit && !c->comm_out) FAIL(PMX_E_STATE, "pmx_set_comm_out has not been called");
    const float* A = p.accelerated ? c->Xe[0] : c->X[0];
    const float* St = p.accelerated ? c->Xe[1] : c->X[1];
    switch (phase) {
        case 0:
            rc = phase_mark(c, 0);
            if (rc != PMX_OK) return rc;
            if (!p.use_fixed_steps) {
                rc = enqueue_gram_only(c, A, St, true, true);
                if (rc != PMX_OK) return rc;
            }
            rc = enqueue_grad(c, A, St, 1, 1);
            if (rc == PMX_OK) rc = phase_mark(c, 1);
            if (rc != PMX_OK) return rc;
            rc = c->ssplit ? shard_pack_split(c, 1, !p.use_fixed_steps) : shard_pack(c, 1, !p.use_fixed_steps, 0);
            return rc == PMX_OK ? phase_mark(c, 2) : rc;
        case 1:
            rc = phase_mark(c, 3);
            if (rc == PMX_OK) rc = c->ssplit ? shard_post_split(c, it > 0) : shard_post(c, it > 0);
            if (rc != PMX_OK) return rc;
            if (p.use_fixed_steps) rc = set_fixed_steps(c, p.fixed_steps);
            else {
                rc = shard_gram_in(c);
                if (rc == PMX_OK) rc = enqueue_eig_only(c, true, true, (double)p.step_scale);
            }
            if (rc == PMX_OK) rc = phase_mark(c, 4);
            if (rc != PMX_OK) return rc;
            c->it += 1;
            rc = pgm_enqueue_update(c, true, 0);
            return rc == PMX_OK ? phase_mark(c, 5) : rc;
        case 2: return c->ssplit ? shard_pack_split(c, 0, false) : shard_pack(c, 0, false, 0);
        case 3: return c->ssplit ? shard_post_split(c, 1) : shard_post(c, 1);
        default: FAIL(PMX_E_INVALID, "bad phase %d", phase);
    }
}

static int bsdmm_enqueue_block(pmx_ctx* c, int j, bool gS_from_comm, const float* comm_scalars, int64_t size_global) {
    const pmx_bsdmm_params& p = c->bsd;
    BsdmmArgs u{};
    u.X = c->X[j];
    u.slab = slab_ref(c, j);
    if (gS_from_comm) { u.slab.base = c->comm; u.slab.n = 1; u.slab.ld = (int)c->K; u.slab.extra = nullptr; }
    for (int i = 0; i < p.n_g[j]; ++i) { u.Z[i] = c->Zg[j][i]; u.U[i] = c->Ug[j][i]; u.prox_g[i] = to_dev(p.prox_g[j][i]); }
    u.rows = c->rows[j];
    u.K = (int)c->K;
    u.j = j;
    u.n_g = p.n_g[j];
    u.prox_f = to_dev(p.prox_f[j]);
    u.status = c->dstatus;
    u.partials = c->partials;
    launch_bsdmm_update(u, c->stream);
    (void)comm_scalars; (void)size_global;
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}
static int bsdmm_enqueue_decide(pmx_ctx* c, int j, const float* comm_scalars, int64_t size) {
    const pmx_bsdmm_params& p = c->bsd;
    BsdmmDecideArgs d{};
    d.status = c->dstatus;
    d.partials = c->partials;
    d.j = j;
    d.n_g = p.n_g[j];
    d.size = size;
    d.e_rel = p.e_rel[j];
    d.e_abs = p.e_abs[j];
    d.last_block = j == 1;
    d.comm_scalars = comm_scalars;
    launch_bsdmm_decide(d, c->stream);
    HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_bsdmm_phase(pmx_ctx* c, int phase) {
    if (c) { c->absmax_by_finish = false; c->gram_by_update = false; }
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_BSDMM) FAIL(PMX_E_STATE, "pmx_bsdmm_begin has not been called");
    if (!c->comm) FAIL(PMX_E_STATE, "pmx_set_comm_buffer has not been called");
    const pmx_bsdmm_params& p = c->bsd;
    const float* scal = c->comm + c->N * c->K + (int64_t)c->KP * c->KP + MAXK;
    // the A step is applied before the collective: a repeated iteration would apply it twice on the ranks that did not
    // fault, so the chained K1 (whose faults are repaired by repeating the iteration) is not used here.  (The fp16 kernels' range
    // guard is NOT awaited behind every launch here -- a host sync per K1 of a rank's short iteration --: a refused launch ends the
    // run with chain_error_text's explanation.)
    rc = chain_disable(c);
    if (rc != PMX_OK) return rc;
    if (p.n_order > 0 && !(p.n_order == 2 && p.order[0] == 0 && p.order[1] == 1))
        FAIL(PMX_E_UNSUPPORTED, "row-sharded bsdmm runs the default update order (A, then S) only");
    if (phase == 0) {
        // A step: everything is local (step_A comes from the replicated S)
        rc = phase_mark(c, 0);
        if (rc == PMX_OK) rc = enqueue_steps(c, c->X[0], c->X[1], true, false, 1.0);
        if (rc == PMX_OK) rc = enqueue_grad(c, c->X[0], c->X[1], 1, 0);
        if (rc == PMX_OK) rc = bsdmm_enqueue_block(c, 0, false, nullptr, 0);
        // S step, local part: Gram of the UPDATED A, gS with the updated A, pack
        if (rc == PMX_OK) rc = enqueue_gram_only(c, c->X[0], c->X[1], true, false);
        if (rc == PMX_OK) rc = enqueue_grad(c, c->X[0], c->X[1], 0, 1);
        if (rc == PMX_OK) rc = phase_mark(c, 1);            // (both K1 launches and the A step between them)
        if (rc == PMX_OK) rc = shard_pack(c, 1, true, 4 * p.n_g[0]);
        return rc == PMX_OK ? phase_mark(c, 2) : rc;
    }
    if (phase == 1) {
        rc = phase_mark(c, 3);
        if (rc == PMX_OK) rc = shard_gram_in(c);
        if (rc == PMX_OK) rc = enqueue_eig_only(c, true, false, 1.0);                   // lmax(A^T A) -> step_S
        if (rc == PMX_OK) rc = bsdmm_enqueue_decide(c, 0, scal, c->M_global * c->K);    // block A on the global norms
        if (rc == PMX_OK) rc = phase_mark(c, 4);
        if (rc == PMX_OK) rc = bsdmm_enqueue_block(c, 1, true, nullptr, 0);
        if (rc == PMX_OK) rc = bsdmm_enqueue_decide(c, 1, nullptr, c->N * c->K);
        return rc == PMX_OK ? phase_mark(c, 5) : rc;
    }
    FAIL(PMX_E_INVALID, "bad phase %d", phase);
}

extern "C" int pmx_chain_status(pmx_ctx* c, int* halted, int* reason, int* it_done, int last_tau[2]) {
    if (!c || !halted || !reason || !it_done || !last_tau) FAIL(PMX_E_INVALID, "NULL argument");
    HIP_CHECK(hipSetDevice(c->device));
    int rc = read_status(c);
    if (rc != PMX_OK) return rc;
    *halted = c->hstatus->halt;
    *reason = c->hstatus->reason;
    *it_done = c->hstatus->it_done;
    if (c->hstatus->tail_fault)    // its update of iteration it_done was not applied here, but the other ranks applied theirs
        FAIL(PMX_E_STATE, "the fused adaprox tail (k_ada_tail) found its workgroups not co-resident in a row-sharded run: "
                          "this GPU is shared with other work; set PMX_TAIL_FUSED=0");
    // The drivers re-enqueue from it_done after a repaired halt: the host-side iteration counter and Nesterov sequence ran
    // ahead with the iterations of the chunk that were skipped on the device (pmx_pgm_phase(1) draws one omega per ENQUEUED
    // iteration), so they are rewound exactly as pmx_pgm_run does after a chain fault.
    auto rewind_host_sequence = [&]() {
        c->it = c->hstatus->it_done;
        if (c->algo == ALG_PGM && c->pgm.accelerated) {
            c->nest_t = 1.0;
            for (int i = 0; i <= c->it; ++i) c->nest_t = 0.5 * (1.0 + sqrt(4.0 * c->nest_t * c->nest_t + 1.0));
        }
    };
    if (c->hstatus->k1_fault) {    // nothing of iteration it_done was applied on any rank (collective halt flag): fall back, retry
        int again = 0;
        rc = chain_fault_fallback(c, &again);
        if (rc != PMX_OK) return rc;
        rewind_host_sequence();
        *halted = 1;
        *reason = HALT_RETRY;
    } else if (c->hstatus->halt && c->hstatus->reason == HALT_PEER) {
        rc = clear_halt(c);
        if (rc != PMX_OK) return rc;
        HIP_CHECK(hipStreamSynchronize(c->stream));
        c->hstatus->halt = 0;
        c->absmax_by_finish = false;
        rewind_host_sequence();
    }
    last_tau[0] = c->hstatus->last_tau[0];
    last_tau[1] = c->hstatus->last_tau[1];
    return PMX_OK;
}

extern "C" int pmx_adaprox_more_subs(pmx_ctx* c, int t0, int n) {
    int rc = require_ready(c);
    if (rc != PMX_OK) return rc;
    if (c->algo != ALG_ADAPROX) FAIL(PMX_E_STATE, "pmx_adaprox_begin has not been called");
    rc = clear_halt(c);
    if (rc != PMX_OK) return rc;
    // Launches cover whole groups of `nt` passes, so the count really enqueued for the halted iteration is kept here
    // (the caller's t0 is its own, un-rounded tally), per iteration: the caller re-enqueues the iterations that
    // followed the halted one before it learns that one of them halted in turn.
    (void)t0;
    const int it_halted = c->hstatus->it_done;           // refreshed by the pmx_chain_status call that reported the halt
    pmx_ctx::SubRec& r = c->sub_rec[it_halted & 63];
    if (r.it != it_halted) FAIL(PMX_E_STATE, "iteration %d has not been enqueued by pmx_adaprox_phase(1)", it_halted);
    c->sub_nt = r.nt;
    const int t_to = ada_enqueue_subs(c, r.enq, n);
    r.enq = t_to;
    rc = ada_enqueue_tail(c, t_to);
    c->absmax_by_finish = rc == PMX_OK;
    return rc;
}

extern "C" int pmx_iter_result(pmx_ctx* c, pmx_result* res) {
    if (!c || !res) FAIL(PMX_E_INVALID, "NULL argument");
    int rc = read_status(c);
    if (rc != PMX_OK) return rc;
    fill_result(c, res, 0);
    return PMX_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Collectives through the C ABI (missing 7): the row-sharded protocol (pmx_*_phase) leaves its one
// all-reduce (or reduce-scatter + all-gather) per iteration to the caller; a caller without torch.distributed gets them
// here: RCCL itself, looked up at run time (dlopen: the library a process has loaded already -- torch ships its own -- or
// librccl.so.1 of the ROCm installation; PMX_RCCL_LIB overrides), one communicator per context, every collective on the
// context's stream, i.e. ordered with its kernels without a single host synchronisation.  The 128-byte id of rank 0
// reaches the other ranks by whatever the caller bootstraps with (MPI, a file, torch.distributed's store).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct RcclId { char internal[128]; };                    // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;             // ncclFloat32, ncclSum
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.handle) return PMX_OK;
    const char* names[] = {getenv("PMX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)                            // first what the process has loaded already (one RCCL per process)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h && !getenv("PMX_RCCL_LIB")) {
        // An RCCL under another name may be mapped already (torch ships its own): loading the system library beside it would
        // bring a second HIP runtime into the process (_lib.py).  Look through the loaded objects before loading anything.
        struct Probe { char path[512]; } probe{};
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) {
            const char* nm = info->dlpi_name;
            if (nm && strstr(nm, "librccl") != nullptr) { snprintf(static_cast<Probe*>(data)->path, sizeof(Probe::path), "%s", nm); return 1; }
            return 0;
        }, &probe);
        if (probe.path[0]) {
            h = dlopen(probe.path, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (!h) FAIL(PMX_E_UNSUPPORTED, "an RCCL is mapped already (%s) but cannot be opened; set PMX_RCCL_LIB to the library to use", probe.path);
        }
    }
    for (const char* n : names)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* why = dlerror();
        FAIL(PMX_E_UNSUPPORTED, "RCCL not found (librccl.so.1; set PMX_RCCL_LIB): %s", why ? why : "no further detail from the loader");
    }
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.ReduceScatter || !a.AllGather || !a.GetErrorString)
        FAIL(PMX_E_UNSUPPORTED, "the RCCL library found lacks an entry point this layer needs");
    g_rccl = a;
    return PMX_OK;
}
}  // namespace
#define RCCL_CHECK(expr)                                                                              \
    do {                                                                                              \
        const int r_ = (expr);                                                                        \
        if (r_ != 0) FAIL(PMX_E_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_));                     \
    } while (0)

extern "C" int pmx_comm_unique_id(unsigned char id[128]) {
    if (!id) FAIL(PMX_E_INVALID, "NULL argument");
    int rc = rccl_load();
    if (rc != PMX_OK) return rc;
    RcclId u;
    RCCL_CHECK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, 128);
    return PMX_OK;
}
extern "C" int pmx_comm_init(pmx_ctx* c, const unsigned char id[128], int rank, int world) {
    if (!c || !id) FAIL(PMX_E_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) FAIL(PMX_E_INVALID, "bad rank/world");
    if (c->rccl_comm) FAIL(PMX_E_STATE, "this context has a communicator already");
    int rc = rccl_load();
    if (rc != PMX_OK) return rc;
    HIP_CHECK(hipSetDevice(c->device));
    RcclId u;
    memcpy(u.internal, id, 128);
    RCCL_CHECK(g_rccl.CommInitRank(&c->rccl_comm, world, u, rank));
    c->rccl_rank = rank; c->rccl_world = world;
    return PMX_OK;
}
extern "C" int pmx_comm_destroy(pmx_ctx* c) {
    if (!c || !c->rccl_comm) return PMX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    const int r = g_rccl.CommDestroy(c->rccl_comm);
    c->rccl_comm = nullptr;
    if (r != 0) FAIL(PMX_E_HIP, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    return PMX_OK;
}
static int comm_ready(pmx_ctx* c) {
    if (!c) FAIL(PMX_E_INVALID, "ctx is NULL");
    if (!c->rccl_comm) FAIL(PMX_E_STATE, "pmx_comm_init has not been called");
    HIP_CHECK(hipSetDevice(c->device));
    return PMX_OK;
}
extern "C" int pmx_comm_all_reduce(pmx_ctx* c, float* dptr, int64_t count) {
    int rc = comm_ready(c);
    if (rc != PMX_OK) return rc;
    if (!dptr || count < 0) FAIL(PMX_E_INVALID, "bad argument");
    RCCL_CHECK(g_rccl.AllReduce(dptr, dptr, (size_t)count, RCCL_FLOAT32, RCCL_SUM, c->rccl_comm, c->stream));
    return PMX_OK;
}
extern "C" int pmx_comm_reduce_scatter(pmx_ctx* c, const float* dsend, float* drecv, int64_t recvcount) {
    int rc = comm_ready(c);
    if (rc != PMX_OK) return rc;
    if (!dsend || !drecv || recvcount < 0) FAIL(PMX_E_INVALID, "bad argument");
    RCCL_CHECK(g_rccl.ReduceScatter(dsend, drecv, (size_t)recvcount, RCCL_FLOAT32, RCCL_SUM, c->rccl_comm, c->stream));
    return PMX_OK;
}
extern "C" int pmx_comm_all_gather(pmx_ctx* c, const float* dsend, float* drecv, int64_t sendcount) {
    int rc = comm_ready(c);
    if (rc != PMX_OK) return rc;
    if (!dsend || !drecv || sendcount < 0) FAIL(PMX_E_INVALID, "bad argument");
    RCCL_CHECK(g_rccl.AllGather(dsend, drecv, (size_t)sendcount, RCCL_FLOAT32, c->rccl_comm, c->stream));
    return PMX_OK;
}
