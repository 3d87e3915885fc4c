/*
 * pmx.h -- C ABI of libpmx.so, the MI355X (gfx950) engine behind proxmin_amd.nmf.nmf().
 *
 * The reference (pmelchior/proxmin) has no FFI: its boundary is Python callables.  This header
 * is the boundary a maintainer of the reference would bind with ctypes to move the
 * `proxmin.nmf.nmf()` path onto the GPU (see INTEGRATION.md for the stub).  Every entry point
 * names the reference code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C types only; device memory is owned by the context unless stated otherwise;
 *   - every function returns PMX_OK (0) or a negative PMX_E_* code; pmx_last_error() gives text;
 *   - factors are float32 on the device.  A is M x K row-major.  S (K x N in the reference) is
 *     held TRANSPOSED as St = S^T, N x K row-major, so that both blocks are tall matrices with
 *     K contiguous components per row (host wrappers transpose on upload/download);
 *   - "block" j: 0 = A, 1 = S, exactly the order of X = [A, S] in proxmin/nmf.py:147;
 *   - all launches go to the context's stream; calls that return results synchronise it.
 */
#ifndef PMX_H
#define PMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_ABI_VERSION 5

/* ---- status codes -------------------------------------------------------------------- */
enum {
    PMX_OK = 0,
    PMX_E_INVALID = -1,     /* bad argument (the reference would `assert`)                   */
    PMX_E_HIP = -2,         /* a HIP runtime call failed                                      */
    PMX_E_NOMEM = -3,
    PMX_E_UNSUPPORTED = -4, /* valid in the reference but not implemented on the device       */
    PMX_E_STATE = -5        /* call order (e.g. run before Y/factors were set)                */
};

/* ---- Y storage / contraction arithmetic ---------------------------------------------- */
enum {
    PMX_MODE_F32 = 0,   /* Y fp32 in HBM, A@S / R@S^T / A^T@R on v_mfma_f32_32x32x2_f32 (exact f32) */
    /* (1 was a plain-bf16 mode -- Y rounded to bf16 in HBM -- that was never built: 2^-9 relative error on Y cannot meet
       the path's rtol 1e-4; the split modes below keep Y in fp32)                                                       */
    PMX_MODE_BF16X3 = 2, /* Y fp32 in HBM, operands split into bf16 terms (3 for A@S, 2 for the gradients), fp32 accumulate */
    PMX_MODE_F16X2 = 3,  /* as BF16X3, but K = 64 / M % 128 = 0 / N % 256 = 0 shapes and K = 128 / M % 128 = 0 / N % 128 = 0
                            shapes (the latter without weights) run the two-term fp16 kernels (power-of-two operand
                            scales from the factor maxima; 9 instead of 12 MFMA products per MAC).  The residual is scaled
                            by ONE power of two from the bound max|Y| + K max|A| max|S|: a gradient launch that finds
                            K max|A| max|S| > 2^16 max|Y| (factors that ran away from the data: entries with A S = 0 would
                            fall out of fp16's range) is refused before anything is written, and the context repeats the
                            iteration with the exact-f32 kernel of its frame and keeps it (pmx_k1_info reports both);
                            entry points that run one iteration per call await every such launch and switch on the spot;
                            row-sharded bSDMM alone returns PMX_E_HIP with that explanation. */
    PMX_MODE_F16X2R = 5, /* PMX_MODE_F16X2 with the RESIDUAL in exact fp32's error class (the two-term product carries the operands' 2^-23
                            representation errors COHERENTLY into the gradients: twice fp32's gradient error).
                            [r5] K1's K = 64 and 128, no weights (k_grad_f16_v8<HH>, k_grad_f16_k128<HH>; pmx_k1_info kernels 11, 12): the
                            residual comes from the HIGH x HIGH product alone, P0 = a0 s0 (one MFMA product instead of three), and what that
                            leaves out is restored exactly through K x K matrices, A S - a0 s0 = A s_r + a_r s0 (x_r = X - x0):
                            gA += A (s_r S^T) + a_r (s0 S^T), gS += (A^T a_r) S + (A^T a0) s_r -- three small launches beside K1
                            (k_gfix.hip) whose result the update kernels fold like one more gradient slab.  7 instead of 9 MFMA products per
                            MAC, no representation error left in the residual at all.
                            [r4] K1's K = 32 (k_grad_f16_k32<R3>, kernel 10), the loss-only passes and PMX_F16_R3=1 (k_grad_f16_v8<R3>,
                            kernel 9): A S takes the third fp16 terms of A and S (ah s3 + a3 sh beside ah sl + al sh) and keeps everything
                            but ah sh in a second accumulator, R = (P_hh - Y) + P_lo -- 5 instead of 3 products in that contraction.
                            Weighted contexts: the kernels of PMX_MODE_F16X2 (the missing term W o (A s_r + a_r s0) does not factor). */
    PMX_MODE_F64 = 4,    /* [ABI v3] fp64 operands, products and sums -- what the reference computes for fp64 inputs
                            (nmf.py:39-41 keeps the dtype of its arguments).  The fused loops of the three back-ends with this
                            library's operators and step rules: pmx_set_Y_host_f64, pmx_upload_f64 / pmx_download_f64,
                            pmx_pgm_begin / _run (pgm, FISTA; [r6] with the Beck-Teboulle line search on the matrix-core kernels), pmx_adaprox_begin / _run
                            (all six schemes, warm start, constant steps), pmx_bsdmm_begin / _run, pmx_grad, pmx_loglike, pmx_step_pgm,
                            pmx_iter_result; every other entry point (Barzilai-Borwein, the *_split pieces for user callables,
                            sharding) returns PMX_E_UNSUPPORTED in such a context -- the host wrappers compute those cases in
                            fp32 (and say so).  Two sets of kernels behind the same entry points: SMALL problems (K <= 16,
                            M N <= 2^20, M, N <= 8192: the reference's own examples and BASELINE cfg1) run fused launches of
                            plain fp64 FMAs (k_small_f64.hip); everything else up to K = 128 -- [r6] -- runs one
                            v_mfma_f64_16x16x4_f64 pass per gradient and the updates as plain launches (k_grad_f64.hip,
                            k_big_f64.hip; PMX_F64_BIG=0 in the environment restores the fp32 computation of those shapes),
                            which also take WEIGHTS (pmx_set_W_host_f64) */
    PMX_MODE_F64_MFMA = 6 /* [ABI v5] PMX_MODE_F64 on the matrix-core kernels whatever the shape (a weighted likelihood or the line search on a
                            small problem; tests) */
};

/* ---- proximal operators: proxmin/operators.py:20-160 ---------------------------------- */
enum {
    PMX_PROX_NONE = -1,      /* prox=None: adaprox skips the sub-iteration loop (algorithms.py:380) */
    PMX_PROX_ID = 0,         /* prox_id          operators.py:20  */
    PMX_PROX_ZERO = 1,       /* prox_zero        operators.py:26  */
    PMX_PROX_PLUS = 2,       /* prox_plus        operators.py:33  */
    PMX_PROX_UNITY = 3,      /* prox_unity       operators.py:41  (axis given in pmx_prox.unit)      */
    PMX_PROX_UNITY_PLUS = 4, /* prox_unity_plus  operators.py:48  */
    PMX_PROX_MIN = 5,        /* prox_min         operators.py:55  */
    PMX_PROX_MAX = 6,        /* prox_max         operators.py:71  */
    PMX_PROX_HARD = 7,       /* prox_hard        operators.py:109 */
    PMX_PROX_HARD_PLUS = 8,  /* prox_hard_plus   operators.py:127 */
    PMX_PROX_SOFT = 9,       /* prox_soft        operators.py:138 */
    PMX_PROX_SOFT_PLUS = 10  /* prox_soft_plus   operators.py:153 */
};

/* One proximal operator.  `unit` says which sum prox_unity* normalises by, in DEVICE layout:
 * 0 = along the K components of one row (numpy axis=1 for A, axis=0 for S),
 * 1 = along the rows, per component     (numpy axis=0 for A, axis=1 for S).
 * `relative` = 1 is the reference's type="relative" (threshold multiplied by the step the
 * solver passes to the prox, operators.py:4-14), 0 is type="absolute". */
typedef struct pmx_prox {
    int32_t op;
    int32_t unit;
    double thresh;      /* [ABI v3] fp64 (was float): the fp64 contexts threshold with the caller's own value -- 0.01f differs from
                           0.01 by 2e-8 relative; the fp32 kernels use (float)thresh exactly as before */
    int32_t relative;
    int32_t reserved;   /* 0 */
} pmx_prox;

#define PMX_MAX_SEQ 4 /* AlternatingProjections of up to 4 built-ins (operators.py:187-211) */
#define PMX_MAX_G 4   /* constraints per block for bsdmm                                      */

/* A (possibly composite) operator: seq[0..n-1] applied in the order stored, `repeat` times.
 * (The host wrapper reverses AlternatingProjections' list, which applies last-to-first.) */
typedef struct pmx_proxseq {
    int32_t n;      /* 0 => PMX_PROX_NONE */
    int32_t repeat; /* >= 1 */
    pmx_prox seq[PMX_MAX_SEQ];
} pmx_proxseq;

/* ---- adaprox moment schemes: proxmin/algorithms.py:147-245 ----------------------------- */
enum { PMX_ADAM = 0, PMX_NADAM = 1, PMX_AMSGRAD = 2, PMX_PADAM = 3, PMX_ADAMX = 4, PMX_RADAM = 5 };

/* ---- device buffers addressable through pmx_upload / pmx_download ---------------------- */
enum {
    PMX_BUF_A = 0,   /* M x K                         */
    PMX_BUF_ST = 1,  /* N x K  (S transposed)         */
    PMX_BUF_GA = 2,  /* last gradient wrt A, M x K    (pgm return value, algorithms.py:144) */
    PMX_BUF_GST = 3, /* last gradient wrt S^T, N x K  */
    PMX_BUF_MA = 4, PMX_BUF_MST = 5,   /* adaprox first moments  (algorithms.py:348-349)  */
    PMX_BUF_VA = 6, PMX_BUF_VST = 7,   /* adaprox second moments (algorithms.py:352-353)  */
    PMX_BUF_VHA = 8, PMX_BUF_VHST = 9, /* adaprox Vhat (only when warm-started, :356-359) */
    PMX_BUF_EVAL_A = 10, PMX_BUF_EVAL_ST = 11, /* pgm: the point the gradient / step were evaluated at (_X, algorithms.py:93-99) */
    PMX_BUF_TMP_A = 12, PMX_BUF_TMP_ST = 13,   /* host round trip of a user-defined prox: its argument, then its result       */
    PMX_BUF_PSI_A = 14, PMX_BUF_PSI_ST = 15,   /* adaprox Psi of the current iteration (algorithms.py:375-377)                */
    PMX_BUF_Z0 = 16, /* + block*PMX_MAX_G + i : bsdmm Z_i of block (utils.py:244-254)     */
    PMX_BUF_U0 = 32, /* + block*PMX_MAX_G + i : bsdmm U_i                                 */
    PMX_BUF_TG0 = 48, /* + block*PMX_MAX_G + i : host round trip of a user-defined proxs_g member: its argument, then its result */
    PMX_BUF_STEP_A = 64, PMX_BUF_STEP_ST = 65, /* pgm: per-element steps of a user `step` that returned arrays (pmx_pgm_step_arrays) */
    PMX_BUF_BT_A = 66, PMX_BUF_BT_ST = 67      /* pgm line search with a user-defined prox: its argument, then its result (pmx_pgm_bt_split) */
};

typedef struct pmx_ctx pmx_ctx;

/* ---- library ---------------------------------------------------------------------------- */
int pmx_abi_version(void);
const char* pmx_last_error(void);
/* number of visible HIP devices (0 when there is no GPU; never fails) */
int pmx_device_count(void);
/* sizeof of the parameter / result structs below, in this order: pmx_proxseq, pmx_pgm_params, pmx_adaprox_params,
 * pmx_bsdmm_params, pmx_result -- a binding checks its own mirrors against them (tests/test_abi.py) */
int pmx_abi_sizes(int sizes[5]);

/* ---- context ------------------------------------------------------------------------------
 * One context = one nmf() call's device state on one GPU: Y (M x N, this rank's rows), the
 * factors and the solver state.  `stream` is a hipStream_t to launch on (e.g. torch's current
 * stream) or NULL to create a private one.  Replaces the closure set-up of nmf.py:146-148. */
int pmx_ctx_create(pmx_ctx** out, int device, int64_t M, int64_t N, int64_t K, int mode, void* stream);
int pmx_ctx_destroy(pmx_ctx* ctx);
int pmx_ctx_sync(pmx_ctx* ctx);

/* Y: copy from host (row-major float32, leading dimension ld elements) or adopt/convert a
 * device buffer (float32, row-major).  `copy`=0 adopts the fp32 device pointer without copying (caller keeps it
 * alive).  Y is read-only for the solvers (nmf.py:116). */
int pmx_set_Y_host(pmx_ctx* ctx, const float* Y, int64_t ld);
int pmx_set_Y_device(pmx_ctx* ctx, const float* dY, int64_t ld, int copy);
/* K1's frame.  The producer / consumer kernels take M % 128 == 0, N % 256 == 0 (N % 128 at K = 128) and K = 32 / 64 (128 in
 * PMX_MODE_F16X2) only.  A context with another shape works on a ZERO-PADDED frame instead of dropping to the guarded kernels:
 *   rows / columns  M and N rounded up, when that costs at most 25 % more entries: the context keeps its OWN padded copy of Y and W
 *                   (pmx_set_Y_device / pmx_set_W_device with copy = 0 copy anyway: the caller's buffer is not referenced after the
 *                   call; call again after changing it), factor arrays and gradient slabs have the frame's rows;
 *   components      a K between the tuned ones runs the next one's kernel on copies of the factors with that many floats per row
 *                   (made in front of every gradient pass), zero behind column K.
 * frame[0] x frame[1] x frame[2] (= M x N x K when the context is not framed).  Results are those of the M x N x K problem: zero rows /
 * columns / components contribute nothing to the residual, the gradients or the loss; everything outside K1 (update kernels, step
 * rules, collectives) works on the real shape.  PMX_FRAME=0 in the environment switches framing off. */
int pmx_k1_frame(pmx_ctx* ctx, int64_t frame[3]);
/* Weighted likelihood: W is the M x N weight array of nmf.log_likelihood / grad_likelihood (proxmin/nmf.py:13-41),
 * loss = 1/2 sum W (Y - A S)^2, D = W (A S - Y).  Row-major float32, leading dimension ld >= N.  Without a call
 * W == 1 (the reference's default).  A PMX_MODE_F32 context takes weights at any shape; a split-bf16 context where its
 * default kernel applies (K = 64, M % 128 == 0, N % 256 == 0) and fails with PMX_E_UNSUPPORTED elsewhere (the caller then
 * creates an F32 context: proxmin_amd/engine.py:open_weighted).  pmx_set_W_host(ctx, NULL, 0) goes back to W == 1.
 * The default PGM / bSDMM step rule does not exist for an array W (nmf.step_pgm tests `W == 1` on it and raises,
 * nmf.py:63): pmx_pgm_begin without fixed or Barzilai-Borwein steps (and without pmx_pgm_params::unweighted_rule) and
 * pmx_bsdmm_begin fail with PMX_E_INVALID.   */
int pmx_set_W_host(pmx_ctx* ctx, const float* W, int64_t ld);
int pmx_set_W_device(pmx_ctx* ctx, const float* dW, int64_t ld, int copy);

/* raw float32 transfers host <-> one of the PMX_BUF_* arrays (count = number of floats) */
int pmx_upload(pmx_ctx* ctx, int buf, const float* host, int64_t count);
int pmx_download(pmx_ctx* ctx, int buf, float* host, int64_t count);
/* PMX_MODE_F64 contexts: Y (M x N, row pitch ld, copied) and the raw fp64 transfers of PMX_BUF_A / _ST, the adaprox moments
 * PMX_BUF_MA .. _VHST (both directions), PMX_BUF_GA / _GST (download: the gradient pgm returns, algorithms.py:144) and bsdmm's
 * PMX_BUF_Z0 / _U0 + .. (download) */
int pmx_set_Y_host_f64(pmx_ctx* ctx, const double* Y, int64_t ld);
/* [ABI v5] the same from a float64 device buffer (row-major, pitch ld elements; copied: the context keeps its own padded array), and the
 * weights of the likelihood (nmf.py:13-41: M x N, float64, host) of an fp64 context that runs the matrix-core kernels (NULL: back to W == 1;
 * PMX_E_UNSUPPORTED in a context on the small-problem kernels: create it with PMX_MODE_F64_MFMA) */
int pmx_set_Y_device_f64(pmx_ctx* ctx, const double* dY, int64_t ld);
int pmx_set_W_host_f64(pmx_ctx* ctx, const double* W, int64_t ld);
int pmx_upload_f64(pmx_ctx* ctx, int buf, const double* host, int64_t count);
int pmx_download_f64(pmx_ctx* ctx, int buf, double* host, int64_t count);
/* device address of a buffer (for zero-copy interop, e.g. torch.distributed on the comm buffer) */
int pmx_buffer_ptr(pmx_ctx* ctx, int buf, void** dptr, int64_t* count);

/* ---- measurement --------------------------------------------------------------------------
 * With timing on, launches of the fused residual-gradient kernel (K1) are bracketed by HIP events
 * on the context's stream -- every launch for on == 1, every on-th launch for on > 1 (a pair of
 * event records costs several microseconds of stream time, which matters inside a timed loop);
 * pmx_get_timing returns the summed duration and count of the bracketed launches since
 * the last pmx_set_timing(ctx, 1) (synchronises the stream).  bench.py derives roofline.achieved
 * from it. */
int pmx_set_timing(pmx_ctx* ctx, int on);
int pmx_get_timing(pmx_ctx* ctx, double* total_ms, int* launches);
/* Row-sharded runs (pmx_*_phase): a per-phase timeline of a rank's iteration from HIP events on the context's stream, every
 * `every`-th iteration (0: off).  pmx_get_phase_timing returns the mean duration in ms of
 *   [0] phase 0 up to and including K1 (operand maxima / fp16 split of A / Gram launches that precede it included)
 *   [1] the rest of phase 0 (k_shard_pack*: gS slabs folded into the collective's buffer)
 *   [2] end of phase 0 -> start of phase 1: the collective the caller enqueued in between (all-reduce / reduce-scatter)
 *   [3] phase 1 up to the update (k_shard_post*, the step rule of the sharded pgm / bsdmm)
 *   [4] the update (fused adaprox tail, or the kernels it stands for; pgm / bsdmm block updates)
 *   [5] end of phase 1 -> start of the next phase 0 (S-split: the all-gather of S; else ~0)
 * and the number of iterations averaged (synchronises the stream).  bench.py --gpus N prints it as "phases_ms" so that a
 * multi-GPU line explains itself. */
int pmx_set_phase_timing(pmx_ctx* ctx, int every);
int pmx_get_phase_timing(pmx_ctx* ctx, double ms[6], int* iterations);
/* average duration (ms, HIP events) of `reps` back-to-back launches of K1 at the current factors with
 * only the requested outputs (do_A / do_S; 0,0 = residual + loss only): kernel ablation for tuning. */
int pmx_time_grad(pmx_ctx* ctx, int do_A, int do_S, int reps, double* avg_ms);

/* How the fused residual-gradient kernel (K1) of this context is laid out -- for tests and bench.py, which must be able
 * to tell which implementation produced a number: info[0] = kernel (0 exact-fp32 MFMA k_grad_f32, 1 split-bf16
 * k_grad_bf16*, 2 two-term fp16 k_grad_f16_v8, 4 the small-problem fp32 kernel k_grad_small, 5 two-term fp16 at K = 128
 * k_grad_f16_k128, 6 exact fp32 with producer / consumer wavefronts k_grad_f32_pc, 7 the fp64 small-problem kernels, 8 two-term fp16 at K = 32
 * k_grad_f16_k32, 9 / 10 k_grad_f16_v8<R3> / k_grad_f16_k32<R3>, 11 / 12 k_grad_f16_v8<HH> / k_grad_f16_k128<HH> + the correction slab of k_gfix.hip: PMX_MODE_F16X2R, 13 the fp64 matrix-core passes k64_grad_pass), info[1] = workgroups per gA chain (0: one gA slab per column region,
 * no chains), info[2] / info[3] = gA / gSt slabs the update kernels fold, info[4] x info[5] = row x column regions
 * (= workgroups), info[6] = row panels per region, info[7] = times this context left the chained mode after a fault
 * + 1000 x times it left the fused adaprox tail (k_ada_tail) + 1000000 if that tail is in use now + 10000000 if a two-term fp16 kernel
 * refused the residual's range and the context went on in exact fp32 (PMX_MODE_F16X2 above; info[0] then names the fp32 kernel). */
int pmx_k1_info(pmx_ctx* ctx, int info[8]);

/* ---- single operations (unit parity tests, and what the reference exposes as functions) --- */
/* nmf.grad_likelihood, W=1 (nmf.py:28-41): gradients at the current A, St into GA / GST.     */
int pmx_grad(pmx_ctx* ctx);
/* nmf.log_likelihood, W=1 (nmf.py:13-25): 1/2 sum (Y - A S)^2 at the current factors.        */
int pmx_loglike(pmx_ctx* ctx, double* out);
/* nmf.step_pgm, W==1 branch (nmf.py:44-65): out[0] = 1/lmax(S S^T), out[1] = 1/lmax(A^T A). */
int pmx_step_pgm(pmx_ctx* ctx, double out[2]);
/* nmf.step_adaprox (nmf.py:91-93): out[0..K) = mean(A,0)/10, out[K..2K) = mean(S,1)/10.      */
int pmx_step_adaprox(pmx_ctx* ctx, float* out);
/* apply one operator to a device buffer in place, as `prox(X, step)` would (operators.py).
 * step_k: K per-component steps (scalar steps: K copies). block selects rows (M or N).        */
int pmx_prox_apply(pmx_ctx* ctx, int buf, const pmx_proxseq* prox, const float* step_k);
/* context-free variant: `prox(X, step)` on a HOST array X (rows x K float32, K <= 128, updated in
 * place): copies to the device, runs the operator kernel, copies back.  This is what calling
 * proxmin.operators.prox_* directly on an ndarray maps to. */
int pmx_prox_array(int device, float* X, int64_t rows, int K, const pmx_proxseq* prox, const float* step_k);
/* [ABI v4] context-free reductions of utils.BarzilaiBorweinStepper.step called as a FUNCTION on host arrays (utils.py:216-241; inside
 * pgm the rule runs fused, k_bb_reduce / k_bb_step): one block's X, G and the previous call's X_prev, G_prev (both NULL in the first
 * call: s = y = 0), `count` elements of float (is_f64 = 0) or double (1).  out = { sum s^2, sum s y, sum y^2, sum G^2, max|X|, max|G| }
 * with s = X - X_prev, y = G - G_prev formed in the arrays' type, products and sums in fp64 in a fixed order. */
int pmx_bb_sums(int device, int is_f64, const void* X, const void* Xprev, const void* G, const void* Gprev, int64_t count, double out[6]);

/* ---- solvers ------------------------------------------------------------------------------
 * Each *_run advances at most `n_iter` iterations from the current device state and returns
 * after the stream is idle.  They can be called repeatedly (the host wrapper calls them with
 * n_iter=1 when a user callback wants to see every iterate, algorithms.py:90/368/802).       */

typedef struct pmx_pgm_params { /* algorithms.pgm arguments, algorithms.py:12-23 */
    pmx_proxseq prox[2];
    int32_t accelerated;  /* Nesterov/FISTA extrapolation (utils.py:193-206) */
    float step_scale;     /* multiplies the Lipschitz steps of nmf.step_pgm (1 = reference default) */
    int32_t use_fixed_steps; /* 1: use fixed_steps[] instead of the Lipschitz rule (user `step`) */
    int32_t unweighted_rule; /* (in what used to be padding) 1: the caller MEANS the unweighted Lipschitz rule although the context carries
                                weights -- the reference's `step_pgm(*X)` called without its W argument, e.g. inside a user lambda;
                                0: pmx_pgm_begin refuses the rule on a weighted context like `partial(step_pgm, W=W)` does (nmf.py:63,152) */
    double fixed_steps[2];
    double e_rel[2];      /* algorithms.py:66-68 */
    int32_t bb_type;      /* 0: off; 1 / 2: utils.BarzilaiBorweinStepper(type) as the step rule (utils.py:209-241) */
    double bb_init_r;     /* its init_r */
    int32_t backtracking; /* 1: Beck-Teboulle line search with f = nmf.log_likelihood (algorithms.py:110-127) */
    int32_t host_prox[2]; /* 1: prox of block j is a user-defined Python callable, applied by the host between the two
                             halves of pmx_pgm_split (prox[j] is ignored)                                          */
} pmx_pgm_params;

typedef struct pmx_result {
    int32_t iterations;   /* iterations executed by THIS call                                 */
    int32_t total_iterations; /* since pmx_*_begin                                             */
    int32_t stopped;      /* 1 if the convergence test ended the run (algorithms.py:134)      */
    int32_t converged[2]; /* last evaluated per-block test                                    */
    double steps[2];      /* pgm/bsdmm: last step sizes (algorithms.py:144)                   */
    int64_t sub_iterations[2]; /* adaprox: accumulated proximal sub-iterations (:398)         */
} pmx_result;

int pmx_pgm_begin(pmx_ctx* ctx, const pmx_pgm_params* p);
int pmx_pgm_run(pmx_ctx* ctx, int n_iter, pmx_result* res);
/* ONE iteration in pieces, for the callbacks the reference takes as Python callables (algorithms.py:37-39, 73-77, 105-108)
 * when they are not objects of this library -- everything but the callable itself stays on the device:
 *   phase 0  gradient at the evaluation point (and the Lipschitz steps unless the caller supplies steps): afterwards
 *            PMX_BUF_EVAL_* and PMX_BUF_GA / GST hold what `step(*_X, it=it, grads=G)` is called with;
 *   phase 1  (only with host_prox) T_j = _X_j - s_j G_j into PMX_BUF_TMP_* for the blocks with a user prox: the caller
 *            downloads it, applies `prox(T_j, s_j)`, uploads the result into the same buffer;
 *   phase 2  the update (built-in operators fused as always), extrapolation, stopping test; fills *res.
 * `steps` (phases 1, 2): the two step sizes a user `step` returned, or NULL to keep the device's.  Lipschitz / fixed /
 * user / Barzilai-Borwein steps (the latter evaluated on the device in phase 0, as in a fused iteration); not with
 * backtracking (phase 0 alone is allowed there: see pmx_pgm_set_fixed_steps). */
int pmx_pgm_split(pmx_ctx* ctx, int phase, const double* steps, pmx_result* res);
/* [ABI v3] ONE iteration of pgm WITH the Beck-Teboulle line search (algorithms.py:110-127) when a block's prox is a user callable
 * (pmx_pgm_params::host_prox): every trial of that block takes a host round trip, everything else stays on the device.
 *   phase 0  start the iteration.  Returns with *need = bit mask of the blocks whose prox the caller now owes: their arguments
 *            _X_j - T_j s_j G_j are in PMX_BUF_BT_A / _ST, eff_steps[j] = T_j s_j is the step the reference passes the callable
 *            (:108, :125); the caller applies it, uploads the result into the same buffer and calls
 *   phase 1  adopt the results, evaluate the sufficient-decrease test; a failed test halves T of the block with the largest
 *            relative update and -- if that block's prox is the caller's -- returns with *need set again.
 * *need == 0: the iteration is complete and *res is filled.  Steps: the device's Lipschitz rule, or constants
 * (pmx_pgm_set_fixed_steps before phase 0: a user `step`). */
int pmx_pgm_bt_split(pmx_ctx* ctx, int phase, int* need, double eff_steps[2], pmx_result* res);
/* A user `step` may return ARRAYS that broadcast against the blocks (algorithms.py:106-108: `_X[j] - S[j] * G[j]`,
 * `prox[j](.., S[j])`): the caller broadcasts block j's to rows x K (S: N x K, transposed like everything of S), uploads it
 * into PMX_BUF_STEP_A / _ST and sets bit j of `mask`; phases 1 and 2 of pmx_pgm_split then take that block's step from the
 * buffer, element by element (the proximal operators get it per element as well), until the mask is cleared. */
int pmx_pgm_step_arrays(pmx_ctx* ctx, int mask);
/* a context begun with use_fixed_steps: new constants for the iterations that follow -- a user `step` evaluated on the host
 * once per iteration next to the device's backtracking line search (algorithms.py:106 with :110-127), which pmx_pgm_split
 * does not do: the caller sets the steps, then runs pmx_pgm_run(ctx, 1, ..) */
int pmx_pgm_set_fixed_steps(pmx_ctx* ctx, const double steps[2]);

typedef struct pmx_adaprox_params { /* algorithms.adaprox arguments, algorithms.py:248-265 */
    pmx_proxseq prox[2];
    int32_t scheme;
    double b2, eps, p;
    int32_t check_convergence;
    int32_t prox_max_iter;
    int32_t warm_vhat;    /* 1: Vhat buffers were uploaded (true AMSGrad/PAdam/AdamX running max) */
    int32_t use_fixed_steps; /* 1: alpha = fixed_alpha (user `step` returning constants); 2: a user `step` callable:
                                the caller writes the per-component steps with pmx_adaprox_set_alpha before every iteration */
    double fixed_alpha[2];
    double e_rel[2];
    int32_t host_prox[2];   /* 1: prox of block j is a user-defined Python callable: its proximal loop (algorithms.py:383-400)
                               runs around the callable on the host between the halves of pmx_adaprox_split           */
} pmx_adaprox_params;

int pmx_adaprox_begin(pmx_ctx* ctx, const pmx_adaprox_params* p, int warm_moments);
/* b1: n_iter values b1[it] for the iterations of this call (algorithms.py:327-330);
 * b1_prev: b1[it-1] for the first of them (adamx, :213; the reference reads b1[-1] at it=0). */
int pmx_adaprox_run(pmx_ctx* ctx, int n_iter, const double* b1, double b1_prev, pmx_result* res);
/* per-component step sizes of the next iteration (use_fixed_steps == 2): alpha[0..K) for A, alpha[K..2K) for S -- what a
 * user `step(*X, it=it)` returned (algorithms.py:370), broadcast to components by the caller */
int pmx_adaprox_set_alpha(pmx_ctx* ctx, const float* alpha);
/* ONE iteration in two halves around the host-side proximal loops of the blocks with host_prox:
 *   phase 0  gradient, moments, X <- X - alpha Phi / Psi (algorithms.py:369-378); maxpsi[j] = max(Psi_j) (:384);
 *            PMX_BUF_A / ST hold the updated blocks, PMX_BUF_PSI_* their Psi;
 *   phase 1  the proximal loops of the other blocks, X <- z, stopping test, next step sizes; host_tau[j] = passes the
 *            host-side loop of block j took (it is only counted); fills *res. */
int pmx_adaprox_split(pmx_ctx* ctx, int phase, int it, double b1_it, double b1_prev, const int* host_tau, double* maxpsi, pmx_result* res);

typedef struct pmx_bsdmm_params { /* algorithms.bsdmm as reachable from nmf(), algorithms.py:653-666 */
    pmx_proxseq prox_f[2];              /* prox_A, prox_S inside prox_f (nmf.py:181-185)       */
    int32_t n_g[2];                     /* constraints per block; 0 = proxs_g[j] is None       */
    pmx_proxseq prox_g[2][PMX_MAX_G];
    double e_rel[2], e_abs[2];
    int32_t n_order;                    /* update_order (algorithms.py:730-736, :805): 0 = the default (A, S); else the  */
    int32_t order[8];                   /* blocks in the order given (a block may be left out or appear more than once)   */
} pmx_bsdmm_params;

int pmx_bsdmm_begin(pmx_ctx* ctx, const pmx_bsdmm_params* p);
int pmx_bsdmm_run(pmx_ctx* ctx, int n_iter, pmx_result* res);
/* One block update of ONE bsdmm iteration in pieces, for user-defined operators (utils.py:307-346 with the callables of
 * proxs_g / prox_f applied by the caller between the pieces; their device slots hold prox_id):
 *   phase 0  step_f of block j (res->steps[j]) and the gradient; host_f: the argument of prox_f -> PMX_BUF_TMP_A / _ST
 *   phase 1  X_j <- prox_f(..) (device operator, or PMX_BUF_TMP_* as the caller left it); no bit set in host_g: the
 *            constraint updates as well; else the arguments X_j + U_i of the user members -> PMX_BUF_TG0 + j*PMX_MAX_G + i
 *   phase 2  (host_g != 0) the constraint updates with Z_i <- PMX_BUF_TG0 + .. for the user members; Boyd's test of the block
 *            (algorithms.py:832-844); last_block: the iteration ends here (counter, stop when all blocks have converged)
 *   step_f_host > 0 (phase 0): the value of a user steps_f_cb for this block instead of the device's Lipschitz rule            */
int pmx_bsdmm_split(pmx_ctx* ctx, int j, int phase, int host_f, unsigned host_g, int last_block, double step_f_host, pmx_result* res);

/* ---- row-sharded multi-GPU --------------------------------------------
 * Rows of Y and A are split over `world` ranks (this context holds M local rows of M_global); S is
 * replicated.  One iteration = phase 0 (local K1 + packing of everything that needs a cross-rank SUM
 * into the comm buffer), ONE all-reduce(sum) of the comm buffer issued by the caller on the same
 * stream (torch.distributed / RCCL over xGMI), phase 1 (update).  The kernels are the ones the
 * single-GPU *_run entry points launch; only the source of gS, of A's column sums and of A's
 * stopping-test sums changes.  The outer stopping test of iteration i is evaluated right after the
 * all-reduce of iteration i+1, before that iteration's update is applied (exact stop semantics).
 *
 * comm layout (float32): [ gSt: N*K | Gram(A): KP*KP | colsum(A): 128 | scalars: 32 ];
 * pmx_comm_layout reports the total count and the three offsets after gSt.  The buffer itself is
 * supplied by the caller (pmx_set_comm_buffer), e.g. a torch tensor, so that the collective library
 * can work on memory it knows. */
/* A user `grad(*X)` callable (algorithms.py:12,248: any differentiable function of the two factors): the caller evaluates it
 * on the host at the point of PMX_BUF_EVAL_A / _ST, uploads the result into PMX_BUF_GA / PMX_BUF_GST, and every solver entry
 * point uses THAT as the gradient of the iteration (the fused residual kernel never runs; the context needs no Y).        */
int pmx_set_host_grad(pmx_ctx* ctx, int on);
int pmx_set_world(pmx_ctx* ctx, int rank, int world, int64_t M_global);
int pmx_comm_layout(pmx_ctx* ctx, int64_t* count, int64_t offsets[3]);
int pmx_set_comm_buffer(pmx_ctx* ctx, float* dptr, int64_t count);
/* S-split: the UPDATE of S sharded as well (adaprox, projection-type prox_S).  An all-reduce is a reduce-scatter followed by
 * an all-gather; here the S update sits between the two: the comm buffer is `world` chunks of
 *     [ gSt rows of rank q (N / world x K) | Gram (KP^2) | colsum(A) (128) | colsum(S) (128) | scalars (32) ],
 * phase 0 packs it, the caller reduce-scatters it (chunk q to rank q, into the pmx_set_comm_out buffer), phase 1 updates A's
 * local rows and S's own N / world columns, and the caller all-gathers the S^T buffer (pmx_buffer_ptr(PMX_BUF_ST)) in place.
 * Call pmx_set_s_split after pmx_set_world and before pmx_adaprox_begin; needs N % world == 0.                               */
int pmx_set_s_split(pmx_ctx* ctx, int on);
int pmx_comm_layout_split(pmx_ctx* ctx, int64_t* count, int64_t* chunk, int64_t offsets[4]);
int pmx_set_comm_out(pmx_ctx* ctx, float* dptr, int64_t count);
/* phase 0: K1 + pack.  phase 1: consume + update (+ `nsub` proximal sub-iteration passes).
 * phase 2: pack only (final stopping-test flush).  phase 3: consume only.  No synchronisation. */
int pmx_adaprox_phase(pmx_ctx* ctx, int phase, int it, double b1_it, double b1_prev, int nsub);
/* pgm: phase 0 = Gram matrices + K1 + pack (gSt, local A^T A, A's stopping sums); phase 1 = global lmax(A^T A),
 * update; phases 2/3 = final stopping-test flush.  Lipschitz or fixed steps (no Barzilai-Borwein, no backtracking). */
int pmx_pgm_phase(pmx_ctx* ctx, int phase, int it);
/* bsdmm: phase 0 = the whole (row-local) A step, then Gram(A_new) + K1 for gS + pack (incl. A's residual norms);
 * phase 1 = global step_S, A's convergence test on the all-reduced norms, S step, close of the iteration.  The
 * single all-reduce sits between the A step and the S step; stop semantics are exact. */
int pmx_bsdmm_phase(pmx_ctx* ctx, int phase);
/* synchronise and report where the chain stands: halted (0/1), reason (1 converged, 2 needs more
 * sub-iteration passes), completed iterations, last tau per block. */
int pmx_chain_status(pmx_ctx* ctx, int* halted, int* reason, int* it_done, int last_tau[2]);
/* resume an iteration that ran out of sub-iteration passes: passes t0 .. t0+n-1, then finish+decide */
int pmx_adaprox_more_subs(pmx_ctx* ctx, int t0, int n);
int pmx_iter_result(pmx_ctx* ctx, pmx_result* res);

/* ---- the collectives themselves, for a caller without torch.distributed -------------------------------------------------
 * No reference counterpart (the reference is single-process).  The phase protocol above leaves its one all-reduce -- or, with
 * the S-split, its reduce-scatter and all-gather -- per iteration to the caller; these entry points are RCCL behind the C ABI:
 * looked up at run time (the RCCL a process has loaded already, else librccl.so.1; PMX_RCCL_LIB overrides), one communicator
 * per context, float32 sums, every call enqueued on the CONTEXT'S STREAM: ordered with the kernels of the phases, no host
 * synchronisation.  Bootstrap: rank 0 calls pmx_comm_unique_id and hands the 128 bytes to the other ranks by its own means
 * (MPI_Bcast, a file, a socket); then every rank calls pmx_comm_init (collective: returns when all `world` ranks have).
 *   iteration, replicated S:   pmx_adaprox_phase(0) ; pmx_comm_all_reduce(comm buffer, pmx_comm_layout count) ; pmx_adaprox_phase(1)
 *   iteration, S-split:        phase(0) ; pmx_comm_reduce_scatter(comm buffer -> comm_out, chunk) ; phase(1) ;
 *                              pmx_comm_all_gather(S^T + rank * N/world * K  ->  S^T, N/world * K)        (in place)
 * pmx_ctx_destroy destroys the communicator. */
int pmx_comm_unique_id(unsigned char id[128]);
int pmx_comm_init(pmx_ctx* ctx, const unsigned char id[128], int rank, int world);
int pmx_comm_all_reduce(pmx_ctx* ctx, float* dptr, int64_t count);                                     /* in place */
int pmx_comm_reduce_scatter(pmx_ctx* ctx, const float* dsend, float* drecv, int64_t recvcount);         /* dsend: world x recvcount */
int pmx_comm_all_gather(pmx_ctx* ctx, const float* dsend, float* drecv, int64_t sendcount);             /* drecv: world x sendcount */
int pmx_comm_destroy(pmx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PMX_H */
