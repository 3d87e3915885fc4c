// fp64 arithmetic at ANY size (K <= 128): the three back-ends of nmf() on fp64 operands, products and sums.
// (included by pmx_api.hip after k_small_f64.hip, whose operator helpers and argument records it uses)
//
// [r6] The reference computes in the dtype of its inputs (nmf.py:39-41) and every example it ships hands it fp64 arrays.
// k_small_f64.hip covers the reference's own examples (K <= 16, M N <= 2^20); above that, fp64 callers were computed in fp32
// and cast back (with a warning since this round).  This file is the path for everything else: the same C entry points
// (a PMX_MODE_F64 context whose shape is outside the small kernels' takes these launches instead), so a caller with fp64
// arrays gets fp64 results at BASELINE's sizes too.
//
//   k64_grad_pass<KP, TRANS, HASW>  (k_grad_f64.hip)  K1 on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s on this part -- the same
//                              as the vector fp64 rate, but with 16 x less operand traffic per FLOP).  ONE gradient per
//                              launch: a workgroup owns 64 rows of the "fixed" factor F (St for gSt, A for gA: its fragments
//                              stay in registers) and sweeps the other factor W in blocks of 64 rows through LDS;
//                                  T(w, f) = sum_k W[w][k] F[f][k] - Y      (nmf.py:39; the residual, one 64 x 64 block)
//                                  gF[f][:] += sum_w T(w, f) W[w][:]        (nmf.py:40-41)
//                              T's accumulator registers ARE the A operand of the second product (contraction over w: the
//                              C/D layout of the f64 instruction puts row q + 4 r of a tile in lanes 16 q .. 16 q + 15 of
//                              register r, which is the A operand's (i = lane & 15, k = lane >> 4) with the contraction index
//                              renamed), so the residual never leaves the registers and nothing is transposed.  gA and gSt are
//                              two launches that each recompute the residual: 8 M N K FLOPs instead of 6 M N K -- the price of
//                              having no cross-workgroup sums, no read-modify-write of partial gradients and no atomics
//                              (fixed summation order, bit-reproducible); bsdmm, which wants one gradient per block update
//                              (nmf.py:181-185), pays nothing.  MFMA-bound: 2 KP instructions of 64 cycles per wave and block
//                              against 8 KB of Y -- the HBM stream is ~20 % of its roof at K = 64.
//   k64_gram_partial<KP> / k64_gram_reduce   the step rule's Gram matrices from fp64 rows (nmf.py:44-65), then k_eig with
//                              EigArgs::force_exact = 2 (power steps on the fp64 matrix: lambda_max to fp64 round-off without the O(K^3) exact solver);
//   k64b_pgm_update<NC>        k64_pgm_update for K <= 128 (a row = half a wave, NC = ceil(K / 32) values per lane);
//   k64b_bt_update / _bt_finish  the Beck-Teboulle line search's trial update and the next evaluation point (host-driven trials);
//   k64b_bsdmm_update<NC>      k64_bsdmm_block spread over the grid + k_bsdmm_decide (the fp32 path's, it only ever saw fp64 sums);
//   k64b_colsum / _alpha / _ada_moment / _ada_sub / _ada_finish / _ada_decide
//                              k64_ada_iter as a chain of launches.  The proximal sub-iteration loop (algorithms.py:380-400)
//                              is one launch per pass; a pass first folds the previous pass's two sums and returns if the loop
//                              has ended (DevStatus::sub_done, sub_tau).  The host enqueues a guess of passes; k64b_ada_finish
//                              halts the chain (HALT_NEED_SUB, nothing written) when the loop needs more -- the host adds
//                              passes and carries on (pmx_api.hip: ada64b_run).  No persistent kernel, no grid barrier: at these
//                              sizes an iteration is milliseconds of MFMA time and the launches are noise.
// Parity: tests/test_gpu_f64_big.py holds all three back-ends to rtol 1e-9 against the fp64 oracle.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// the step rule's Gram matrices from fp64 rows (nmf.py:44-65): fp64 products and sums, fixed order
// ------------------------------------------------------------------------------------------------
constexpr int G64_BLOCKS = 128;
struct Gram64Args {
    const double* X[2];      // factor f: 0 = A (M x K), 1 = St (N x K)
    int64_t rows[2];
    int K;
    double* part;            // [2][G64_BLOCKS][KP*KP]
    double* G;               // [2][KP*KP]
    const DevStatus* status;
    int want[2];
};
template <int KP>
__global__ __launch_bounds__(256) void k64_gram_partial(Gram64Args a) {
    constexpr int TS = KP / 16, CH = 16, NLD = CH * KP / 256;
    __shared__ double xs[CH][KP + 1];
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    if (!a.want[f]) return;
    const int K = a.K;
    const int64_t rows = a.rows[f];
    const double* X = a.X[f];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    double acc[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int jx = 0; jx < TS; ++jx) acc[i][jx] = 0.0;
    const int64_t per = (rows + G64_BLOCKS - 1) / G64_BLOCKS;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = r0 + per < rows ? r0 + per : rows;
    for (int64_t rb = r0; rb < r1; rb += CH) {
        double ld[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = threadIdx.x + u * 256;
            const int rr = e / KP, k = e - rr * KP;
            ld[u] = (rb + rr < r1 && k < K) ? X[(rb + rr) * K + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = threadIdx.x + u * 256;
            xs[e / KP][e % KP] = ld[u];
        }
        __syncthreads();
        const int nrow = r1 - rb < CH ? (int)(r1 - rb) : CH;
        for (int rr = 0; rr < nrow; ++rr) {
            double av[TS], bv[TS];
#pragma unroll
            for (int i = 0; i < TS; ++i) av[i] = xs[rr][ti + 16 * i];
#pragma unroll
            for (int i = 0; i < TS; ++i) bv[i] = xs[rr][tj + 16 * i];
#pragma unroll
            for (int i = 0; i < TS; ++i)
#pragma unroll
                for (int jx = 0; jx < TS; ++jx) acc[i][jx] += av[i] * bv[jx];      // (x_i x_j == x_j x_i and the same row order: symmetric bit for bit)
        }
    }
    double* out = a.part + ((int64_t)f * G64_BLOCKS + blockIdx.x) * KP * KP;       // (a share without rows leaves zeros: the fold reads every slot)
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int jx = 0; jx < TS; ++jx) out[(ti + 16 * i) * KP + tj + 16 * jx] = acc[i][jx];
}
__global__ __launch_bounds__(256) void k64_gram_reduce(Gram64Args a, int KP) {
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    if (!a.want[f]) return;
    const int n = KP * KP, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const double* p = a.part + (int64_t)f * G64_BLOCKS * n + e;
    double s = 0.0;
    for (int b0 = 0; b0 < G64_BLOCKS; b0 += 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p[(int64_t)(b0 + i) * n];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    a.G[(int64_t)f * n + e] = s;
}
void launch_gram64(const Gram64Args& a, int KP, hipStream_t s) {
    const dim3 grid(G64_BLOCKS, 2);
    if (KP == 32) hipLaunchKernelGGL(k64_gram_partial<32>, grid, dim3(256), 0, s, a);
    else if (KP == 64) hipLaunchKernelGGL(k64_gram_partial<64>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k64_gram_partial<128>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k64_gram_reduce, dim3((KP * KP + 255) / 256, 2), dim3(256), 0, s, a, KP);
}

// ------------------------------------------------------------------------------------------------
// rows of up to 128 fp64 values: half a wave per row, value c of lane l32 is component l32 + 32 c
// ------------------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void prox64b_row(double (&v)[NC], const bool (&ok)[NC], const ProxSeq& ps, const double (&sk)[NC]) {
    for (int r = 0; r < ps.repeat; ++r)
        for (int qi = 0; qi < ps.n; ++qi) {
            const pmx_prox& p = ps.seq[qi];
            if (p.op == PMX_PROX_UNITY || p.op == PMX_PROX_UNITY_PLUS) {          // operators.py:41-52 along the K components of the row
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (p.op == PMX_PROX_UNITY_PLUS) v[c] = v[c] < 0.0 ? 0.0 : v[c];
                    s += ok[c] ? v[c] : 0.0;
                }
                s = row_sum_d<32>(s);
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = v[c] / s;                     // (no zero guard, like the reference)
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = prox64_one<32>(v[c], ok[c], p, sk[c]);
            }
        }
}
template <int NC>
__device__ __forceinline__ void fold_slabs64(double (&g)[NC], const bool (&ok)[NC], const double* slab, int nslab, int64_t rows, int K, int64_t r, int l32) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        g[c] = 0.0;
        if (ok[c])
            for (int qs = 0; qs < nslab; ++qs) g[c] += slab[((int64_t)qs * rows + r) * K + l32 + 32 * c];      // fixed order: slab 0, 1, 2, ...
    }
}

// pgm / FISTA update (algorithms.py:93-108,130-135): grid (EW_BLOCKS, 2)
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_pgm_update(Pgm64Args a) {
    __shared__ double scratch[2 * EW_WAVES];
    const int j = blockIdx.y;
    const int halted = __builtin_nontemporal_load(&a.status->halt);
    const double s = a.status->step[j];
    if (halted) return;
    const int64_t rows = a.rows[j];
    const int K = a.K, l32 = threadIdx.x & 31;
    bool ok[NC];
    double sk[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ok[c] = l32 + 32 * c < K; sk[c] = s; }
    double d2 = 0.0, n2 = 0.0;
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    for (int64_t r = hw; r < rows; r += nhw) {
        double g[NC], xo[NC], v[NC];
        fold_slabs64<NC>(g, ok, a.slab[j], a.nslab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            xo[c] = ok[c] ? a.X[j][e] : 0.0;
            const double xe = a.accelerated ? (ok[c] ? a.Xe[j][e] : 0.0) : xo[c];
            v[c] = xe - s * g[c];                                            // algorithms.py:107-108
        }
        prox64b_row<NC>(v, ok, a.prox[j], sk);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!ok[c]) continue;
            const int64_t e = r * K + l32 + 32 * c;
            a.X[j][e] = v[c];
            a.G[j][e] = g[c];
            if (a.accelerated) a.Xe[j][e] = v[c] + a.omega_next * (v[c] - xo[c]);     // algorithms.py:93-95 of the next iteration
            const double d = v[c] - xo[c];
            d2 += d * d;
            n2 += v[c] * v[c];
        }
    }
    double red[2] = {d2, n2};
    block_sum_store<2>(red, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
}
void launch_pgm64b_update(const Pgm64Args& a, hipStream_t s) {
    const dim3 grid(EW_BLOCKS, 2), block(EW_THREADS);
    if (a.K <= 32) hipLaunchKernelGGL(k64b_pgm_update<1>, grid, block, 0, s, a);
    else if (a.K <= 64) hipLaunchKernelGGL(k64b_pgm_update<2>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k64b_pgm_update<4>, grid, block, 0, s, a);
}
// gradient slabs -> G (pmx_grad)
__global__ __launch_bounds__(256) void k64b_fold(Fold64Args a) {
    const int j = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.count[j]; e += (int64_t)gridDim.x * 256) {
        double g = 0.0;
        for (int qs = 0; qs < a.nslab[j]; ++qs) g += a.slab[j][(int64_t)qs * a.count[j] + e];
        a.G[j][e] = g;
    }
}
void launch_fold64b(const Fold64Args& a, hipStream_t s) { hipLaunchKernelGGL(k64b_fold, dim3(1024, 2), dim3(256), 0, s, a); }

// ------------------------------------------------------------------------------------------------
// pgm with the Beck-Teboulle line search (algorithms.py:110-128) in fp64: the trial update of the blocks in `do_block`,
//   X_j = prox_j(E_j - T_j s_j G_j, T_j s_j)                                   (:108, :125)
// and what the test and the choice of the block to shrink need of it: sum (X - X_).G, sum (X - X_)^2, sum X^2, max |G|, max |X_|
// (slots SL_BT0, SL_DIFF2, SL_NORM2, SL_BT0 + 1, SL_BT0 + 2; the host folds them: the loop is data-dependent and every trial ends
// in an evaluation of the likelihood anyway -- pmx_api.hip: pgm64_bt_iteration).  grid (EW_BLOCKS, 2)
// ------------------------------------------------------------------------------------------------
struct Bt64Args {
    double* X[2];
    const double* E[2];      // evaluation point _X (algorithms.py:93-99)
    const double* Xp[2];     // X_ : the iterate before this iteration
    const double* G[2];      // gradient at E (folded)
    int64_t rows[2];
    int K;
    ProxSeq prox[2];
    const DevStatus* status;
    double T[2];
    int do_block[2];
    double* partials;
};
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_bt_update(Bt64Args a) {
    __shared__ double scratch[3 * EW_WAVES];
    __shared__ double smax[2][EW_WAVES];
    const int j = blockIdx.y;
    if (chain_halted(a.status) || !a.do_block[j]) return;
    const double ts = a.T[j] * a.status->step[j];
    const int64_t rows = a.rows[j];
    const int K = a.K, l32 = threadIdx.x & 31;
    bool ok[NC];
    double sk[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ok[c] = l32 + 32 * c < K; sk[c] = ts; }
    double red[3] = {0.0, 0.0, 0.0}, mg = 0.0, mx = 0.0;
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    for (int64_t r = hw; r < rows; r += nhw) {
        double g[NC], xp[NC], v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            g[c] = ok[c] ? a.G[j][e] : 0.0;
            xp[c] = ok[c] ? a.Xp[j][e] : 0.0;
            v[c] = (ok[c] ? a.E[j][e] : 0.0) - ts * g[c];
        }
        prox64b_row<NC>(v, ok, a.prox[j], sk);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!ok[c]) continue;
            a.X[j][r * K + l32 + 32 * c] = v[c];
            const double d = v[c] - xp[c];
            red[0] += d * g[c];
            red[1] += d * d;
            red[2] += v[c] * v[c];
            mg = nanmax(mg, fabs(g[c]));
            mx = nanmax(mx, fabs(xp[c]));
        }
    }
    mg = wave_nanmax(mg);
    mx = wave_nanmax(mx);
    if ((threadIdx.x & 63) == 0) { smax[0][threadIdx.x >> 6] = mg; smax[1][threadIdx.x >> 6] = mx; }
    double r0[1] = {red[0]}, r12[2] = {red[1], red[2]};
    block_sum_store<1>(r0, part_ptr(a.partials, SL_BT0, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    __syncthreads();
    block_sum_store<2>(r12, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    __syncthreads();
    if (threadIdx.x == 0) {
        double m0 = smax[0][0], m1 = smax[1][0];
        for (int qw = 1; qw < EW_WAVES; ++qw) { m0 = nanmax(m0, smax[0][qw]); m1 = nanmax(m1, smax[1][qw]); }
        part_ptr(a.partials, SL_BT0 + 1, j)[blockIdx.x] = m0;
        part_ptr(a.partials, SL_BT0 + 2, j)[blockIdx.x] = m1;
    }
}
// after the line search settled: the next iteration's evaluation point E = X + omega (X - X_) (algorithms.py:93-99; omega = 0: a copy); grid (1024, 2)
struct BtFin64Args { const double* X[2]; const double* Xp[2]; double* E[2]; int64_t count[2]; double omega; const DevStatus* status; };
__global__ __launch_bounds__(256) void k64b_bt_finish(BtFin64Args a) {
    const int j = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.count[j]; e += (int64_t)gridDim.x * 256) {
        const double x = a.X[j][e];
        a.E[j][e] = x + a.omega * (x - a.Xp[j][e]);
    }
}
void launch_bt64_update(const Bt64Args& a, hipStream_t s) {
    const dim3 grid(EW_BLOCKS, 2), block(EW_THREADS);
    if (a.K <= 32) hipLaunchKernelGGL(k64b_bt_update<1>, grid, block, 0, s, a);
    else if (a.K <= 64) hipLaunchKernelGGL(k64b_bt_update<2>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k64b_bt_update<4>, grid, block, 0, s, a);
}
void launch_bt64_finish(const BtFin64Args& a, hipStream_t s) { hipLaunchKernelGGL(k64b_bt_finish, dim3(1024, 2), dim3(256), 0, s, a); }

// ------------------------------------------------------------------------------------------------
// bSDMM block update (algorithms.py:805-844, utils.py:269-391; identity L) over the grid; its sums go to the reduction slots
// k_bsdmm_decide folds (SL_DIFF2, SL_NORM2, SL_G0 + 4 i ..)
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_bsdmm_update(Bsdmm64Args a, double* partials) {
    __shared__ double scratch[4 * PMX_MAX_G * EW_WAVES];
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    const int l32 = threadIdx.x & 31;
    const int K = a.K, j = a.j;
    const double sf = st->step[j];
    const double sg = sf * 1.0 * 2.0 * (double)a.n_g;          // get_step_g (utils.py:269-279), identity L
    const double w = a.n_g > 0 ? sf / sg : 0.0;
    const double nisg = a.n_g > 0 ? -1.0 / sg : 0.0;
    bool ok[NC];
    double skf[NC], skg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ok[c] = l32 + 32 * c < K; skf[c] = sf; skg[c] = sg; }
    double red[2] = {0.0, 0.0}, redg[4 * PMX_MAX_G];
#pragma unroll
    for (int i = 0; i < 4 * PMX_MAX_G; ++i) redg[i] = 0.0;
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    for (int64_t r = hw; r < a.rows; r += nhw) {
        double g[NC], xo[NC], v[NC];
        fold_slabs64<NC>(g, ok, a.slab, a.nslab, a.rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            xo[c] = ok[c] ? a.X[e] : 0.0;
            double dx = 0.0;
            for (int i = 0; i < a.n_g; ++i)                     // utils.py:330-336
                if (ok[c]) dx += w * (xo[c] - a.Z[i][e] + a.U[i][e]);
            v[c] = (xo[c] - dx) - sf * g[c];                    // utils.py:338 + nmf.py:185
        }
        prox64b_row<NC>(v, ok, a.prox_f, skf);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) {
                a.X[r * K + l32 + 32 * c] = v[c];
                const double d = v[c] - xo[c];
                red[0] += d * d;
                red[1] += v[c] * v[c];
            }
        for (int i = 0; i < a.n_g; ++i) {                       // do_the_mm, utils.py:295-304
            double zo[NC], uo[NC], zn[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int64_t e = r * K + l32 + 32 * c;
                zo[c] = ok[c] ? a.Z[i][e] : 0.0;
                uo[c] = ok[c] ? a.U[i][e] : 0.0;
                zn[c] = v[c] + uo[c];
            }
            prox64b_row<NC>(zn, ok, a.prox_g[i], skg);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (ok[c]) {
                    const int64_t e = r * K + l32 + 32 * c;
                    const double rr = v[c] - zn[c], sd = nisg * (zn[c] - zo[c]), un = uo[c] + rr, us = un / sg;
                    a.Z[i][e] = zn[c];
                    a.U[i][e] = un;
                    redg[4 * i + 0] += rr * rr;
                    redg[4 * i + 1] += sd * sd;
                    redg[4 * i + 2] += zn[c] * zn[c];
                    redg[4 * i + 3] += us * us;
                }
        }
    }
    block_sum_store<2>(red, part_ptr(partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
    __syncthreads();
    block_sum_store<4 * PMX_MAX_G>(redg, part_ptr(partials, SL_G0, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
}
void launch_bsdmm64b_update(const Bsdmm64Args& a, double* partials, hipStream_t s) {
    const dim3 grid(EW_BLOCKS), block(EW_THREADS);
    if (a.K <= 32) hipLaunchKernelGGL(k64b_bsdmm_update<1>, grid, block, 0, s, a, partials);
    else if (a.K <= 64) hipLaunchKernelGGL(k64b_bsdmm_update<2>, grid, block, 0, s, a, partials);
    else hipLaunchKernelGGL(k64b_bsdmm_update<4>, grid, block, 0, s, a, partials);
}

// ------------------------------------------------------------------------------------------------
// adaprox (algorithms.py:369-410, nmf.py:91-93) as a chain of launches; Ada64Args as k64_ada_iter, alpha_out = [2][MAXK]
// ------------------------------------------------------------------------------------------------
struct Ada64bArgs {
    Ada64Args a;
    double* partials;        // the context's reduction slots
    double* colpart;         // [2][EW_BLOCKS][MAXK] partial column sums
    int t;                   // k64b_ada_sub: this pass (1-based); k64b_ada_finish: passes enqueued so far for this iteration
};
// column sums of the CURRENT factors (Jacobi: algorithms.py:370 -> nmf.py:93), per workgroup; grid (EW_BLOCKS, 2)
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_colsum(Ada64bArgs b) {
    __shared__ double cs[EW_THREADS / 32][MAXK + 1];
    const Ada64Args& a = b.a;
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y, l32 = threadIdx.x & 31, hwl = threadIdx.x >> 5;
    const int K = a.K;
    const int64_t rows = a.rows[j];
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    double s[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = 0.0;
    for (int64_t r = hw; r < rows; r += nhw)
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (l32 + 32 * c < K) s[c] += a.X[j][r * K + l32 + 32 * c];
#pragma unroll
    for (int c = 0; c < NC; ++c) cs[hwl][l32 + 32 * c] = s[c];
    __syncthreads();
    if (threadIdx.x < MAXK) {
        double tsum = 0.0;
        if ((int)threadIdx.x < 32 * NC)
            for (int qh = 0; qh < EW_THREADS / 32; ++qh) tsum += cs[qh][threadIdx.x];
        b.colpart[((int64_t)j * EW_BLOCKS + blockIdx.x) * MAXK + threadIdx.x] = tsum;
    }
}
// step sizes (nmf.py:93: mean over the rows / 10), or the two constants of a `constant_step`; one workgroup of 256
__global__ __launch_bounds__(256) void k64b_alpha(Ada64bArgs b) {
    const Ada64Args& a = b.a;
    if (chain_halted(a.status)) return;
    const int j = threadIdx.x >> 7, k = threadIdx.x & 127;
    double tsum = 0.0;
    if (!a.use_fixed && k < a.K) {
        const double* p = b.colpart + (int64_t)j * EW_BLOCKS * MAXK + k;
        for (int b0 = 0; b0 < EW_BLOCKS; b0 += 16) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = p[(int64_t)(b0 + i) * MAXK];
#pragma unroll
            for (int i = 0; i < 16; ++i) tsum += v[i];
        }
    }
    a.alpha_out[j * MAXK + k] = k >= a.K ? 0.0 : (a.use_fixed ? a.fixed[j] : (tsum / (double)a.rows[j]) / 10.0);
}
struct Mom64Scalars { double b1, b2, bias1, bias2, rho, rfac, xfac; };
__device__ __forceinline__ Mom64Scalars mom64_scalars(const Ada64Args& a) {
    Mom64Scalars m;
    m.b1 = a.b1t; m.b2 = a.b2;
    const double t = (double)(a.it + 1);
    m.bias1 = 1.0 - pow(m.b1, t); m.bias2 = 1.0 - pow(m.b2, t);
    const double rho_inf = 2.0 / (1.0 - m.b2) - 1.0;
    m.rho = rho_inf - 2.0 * t * pow(m.b2, t) / (1.0 - pow(m.b2, t));
    m.rfac = m.rho > 4.0 ? sqrt((m.rho - 4.0) * (m.rho - 2.0) * rho_inf / (rho_inf - 4.0) / (rho_inf - 2.0) / m.rho) : 1.0;
    m.xfac = ((1.0 - m.b1) * (1.0 - m.b1)) / ((1.0 - a.b1prev) * (1.0 - a.b1prev));
    return m;
}
// moments and update (algorithms.py:375-378), max Psi per workgroup; grid (EW_BLOCKS, 2)
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_ada_moment(Ada64bArgs b) {
    __shared__ double sm[EW_WAVES];
    const Ada64Args& a = b.a;
    if (chain_halted(a.status)) return;
    const int j = blockIdx.y, l32 = threadIdx.x & 31;
    const int K = a.K;
    const int64_t rows = a.rows[j];
    const Mom64Scalars ms = mom64_scalars(a);
    const double b1 = ms.b1, b2 = ms.b2;
    double alpha[NC];
    bool ok[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ok[c] = l32 + 32 * c < K; alpha[c] = ok[c] ? a.alpha_out[j * MAXK + l32 + 32 * c] : 0.0; }
    double maxpsi = -1.0;
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    for (int64_t r = hw; r < rows; r += nhw) {
        double gg[NC];
        fold_slabs64<NC>(gg, ok, a.slab[j], a.nslab[j], rows, K, r, l32);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!ok[c]) continue;
            const int64_t e = r * K + l32 + 32 * c;
            const double g = gg[c];
            const double m = (1.0 - b1) * g + b1 * a.Mm[j][e];
            const double v = (1.0 - b2) * (g * g) + b2 * a.Vv[j][e];
            a.Mm[j][e] = m;
            a.Vv[j][e] = v;
            double phi, psi;
            switch (a.scheme) {
                case PMX_ADAM: phi = m / ms.bias1; psi = sqrt(v / ms.bias2) + a.eps; break;
                case PMX_NADAM: phi = (b1 * m + (1.0 - b1) * g) / ms.bias1; psi = sqrt(v / ms.bias2) + a.eps; break;
                case PMX_RADAM:
                    phi = m / ms.bias1;
                    psi = ms.rho > 4.0 ? sqrt(v / ms.bias2) / ms.rfac : 1.0;
                    if (a.eps > 0.0) psi = fmax(psi, sqrt(a.eps));
                    break;
                default: {   // amsgrad / padam / adamx (algorithms.py:170-221)
                    double cap = v;
                    if (a.Vh[j] != nullptr) {
                        const double old = a.Vh[j][e];
                        cap = fmax(a.scheme == PMX_ADAMX ? ms.xfac * old : old, v);
                        a.Vh[j][e] = cap;
                    }
                    if (a.eps > 0.0) cap = fmax(cap, a.eps);
                    psi = a.scheme == PMX_PADAM ? pow(cap, a.p) : sqrt(cap);
                    phi = m;
                }
            }
            const double xo = a.X[j][e];
            if (a.check_convergence) a.Xp[j][e] = xo;
            const double xn = xo - alpha[c] * phi / psi;
            a.X[j][e] = xn;
            if (a.has_prox[j]) { a.Psi[j][e] = psi; a.z[j][e] = xn; }
            maxpsi = nanmax(maxpsi, psi);
        }
    }
    double mv = wave_nanmax(maxpsi);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mv;
    __syncthreads();
    if (threadIdx.x == 0) {
        double mp = sm[0];
        for (int qw = 1; qw < EW_WAVES; ++qw) mp = nanmax(mp, sm[qw]);
        part_ptr(b.partials, SL_MAXPSI, j)[blockIdx.x] = mp;
    }
}
// one proximal pass (algorithms.py:386-392) of each block whose loop is still running; grid (EW_BLOCKS, 2)
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_ada_sub(Ada64bArgs b) {
    __shared__ double scratch[2 * EW_WAVES];
    const Ada64Args& a = b.a;
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    const int j = blockIdx.y, l32 = threadIdx.x & 31;
    if (!a.has_prox[j]) return;
    if (__builtin_nontemporal_load(&st->sub_done[j])) return;
    const int t = b.t;
    if (t > 1) {             // did pass t - 1 end the loop?  (every wave folds the same 2 x EW_BLOCKS partials in the same order)
        const double d2 = fold_partials(part_ptr(b.partials, SL_SUBR0 + 2 * ((t - 1) % SUB_RING), j), nullptr);
        const double z2 = fold_partials(part_ptr(b.partials, SL_SUBR0 + 2 * ((t - 1) % SUB_RING) + 1, j), nullptr);
        if (d2 <= a.e_rel[j] * a.e_rel[j] * z2) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { st->sub_tau[j] = t - 1; st->sub_done[j] = 1; }
            return;
        }
    }
    const int K = a.K;
    const int64_t rows = a.rows[j];
    double mp = fold_partials_nanmax(part_ptr(b.partials, SL_MAXPSI, j));
    bool ok[NC];
    double gamma[NC], rat[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ok[c] = l32 + 32 * c < K;
        const double al = ok[c] ? a.alpha_out[j * MAXK + l32 + 32 * c] : 0.0;
        gamma[c] = al / mp;                   // :384
        rat[c] = gamma[c] / al;               // NaN if alpha == 0, as in the reference
    }
    double red[2] = {0.0, 0.0};
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    for (int64_t r = hw; r < rows; r += nhw) {       // (whole half-waves take or skip a row: the row sums of prox_unity* need all lanes)
        double zz[NC], v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t e = r * K + l32 + 32 * c;
            zz[c] = ok[c] ? a.z[j][e] : 0.0;
            const double x = ok[c] ? a.X[j][e] : 0.0, ps = ok[c] ? a.Psi[j][e] : 0.0;
            v[c] = zz[c] - rat[c] * ps * (zz[c] - x);
        }
        prox64b_row<NC>(v, ok, a.prox[j], gamma);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ok[c]) {
                const double d = v[c] - zz[c];
                red[0] += d * d;
                red[1] += zz[c] * zz[c];
                a.z[j][r * K + l32 + 32 * c] = v[c];
            }
    }
    block_sum_store<2>(red, part_ptr(b.partials, SL_SUBR0 + 2 * (t % SUB_RING), j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
}
// the loops' verdict after b.t passes; then X <- z (:400) and the sums of the outer test (:403-410); grid (EW_BLOCKS, 2).
// A loop that has neither ended nor used up prox_max_iter halts the chain BEFORE anything is written (HALT_NEED_SUB): the host
// enqueues more passes and this kernel again.
template <int NC>
__global__ __launch_bounds__(EW_THREADS) void k64b_ada_finish(Ada64bArgs b) {
    __shared__ double scratch[2 * EW_WAVES];
    const Ada64Args& a = b.a;
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    const int j = blockIdx.y, l32 = threadIdx.x & 31;
    const int t = b.t;
    bool need = false;
    for (int jj = 0; jj < 2; ++jj) {
        if (!a.has_prox[jj] || t >= a.prox_max_iter) continue;
        if (__builtin_nontemporal_load(&st->sub_done[jj])) continue;
        const double d2 = fold_partials(part_ptr(b.partials, SL_SUBR0 + 2 * (t % SUB_RING), jj), nullptr);
        const double z2 = fold_partials(part_ptr(b.partials, SL_SUBR0 + 2 * (t % SUB_RING) + 1, jj), nullptr);
        if (!(d2 <= a.e_rel[jj] * a.e_rel[jj] * z2)) { need = true; if (blockIdx.x == 0 && j == 0 && threadIdx.x == 0) st->need_sub[jj] = 1; }
    }
    if (need) {
        if (blockIdx.x == 0 && j == 0 && threadIdx.x == 0) {
            st->reason = HALT_NEED_SUB;
            __threadfence();
            st->halt = 1;
        }
        return;
    }
    const int K = a.K;
    const int64_t rows = a.rows[j];
    bool ok[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ok[c] = l32 + 32 * c < K;
    double red[2] = {0.0, 0.0};
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    if (a.has_prox[j] || a.check_convergence)
        for (int64_t r = hw; r < rows; r += nhw)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!ok[c]) continue;
                const int64_t e = r * K + l32 + 32 * c;
                double x;
                if (a.has_prox[j]) { x = a.z[j][e]; a.X[j][e] = x; }
                else x = a.X[j][e];
                if (a.check_convergence) {
                    const double d = x - a.Xp[j][e];
                    red[0] += d * d;
                    red[1] += x * x;
                }
            }
    block_sum_store<2>(red, part_ptr(b.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
}
// outer stopping test and end-of-iteration bookkeeping; one workgroup
__global__ __launch_bounds__(256) void k64b_ada_decide(Ada64bArgs b) {
    const Ada64Args& a = b.a;
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    double d2[2], n2[2];
    for (int j = 0; j < 2; ++j) {
        d2[j] = fold_partials(part_ptr(b.partials, SL_DIFF2, j), nullptr);
        n2[j] = fold_partials(part_ptr(b.partials, SL_NORM2, j), nullptr);
    }
    if (threadIdx.x == 0) {
        int all = 1;
        for (int j = 0; j < 2; ++j) {
            const int tau = a.has_prox[j] ? (st->sub_done[j] ? st->sub_tau[j] : b.t) : 0;
            const int conv = a.check_convergence ? d2[j] <= a.e_rel[j] * a.e_rel[j] * n2[j] : 0;
            all &= conv;
            st->last_tau[j] = tau;
            st->sub_total[j] += tau;
            st->sub_tau[j] = 0;
            st->sub_done[j] = 0;
            st->need_sub[j] = 0;
            st->conv[j] = conv;
            st->norms[j][0] = a.check_convergence ? d2[j] : 0.0;
            st->norms[j][1] = a.check_convergence ? n2[j] : 0.0;
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
#define PMX_ADA64B(KERNEL, GRID, ARGS)                                                              \
    do {                                                                                            \
        if ((ARGS).a.K <= 32) hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(EW_THREADS), 0, s, ARGS);     \
        else if ((ARGS).a.K <= 64) hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(EW_THREADS), 0, s, ARGS); \
        else hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(EW_THREADS), 0, s, ARGS);                      \
    } while (0)
void launch_ada64b_head(const Ada64bArgs& b, hipStream_t s) {        // step sizes, moments, update
    const dim3 grid2(EW_BLOCKS, 2);
    if (!b.a.use_fixed) PMX_ADA64B(k64b_colsum, grid2, b);
    hipLaunchKernelGGL(k64b_alpha, dim3(1), dim3(256), 0, s, b);
    PMX_ADA64B(k64b_ada_moment, grid2, b);
}
void launch_ada64b_sub(const Ada64bArgs& b, hipStream_t s) { const dim3 grid2(EW_BLOCKS, 2); PMX_ADA64B(k64b_ada_sub, grid2, b); }
void launch_ada64b_close(const Ada64bArgs& b, hipStream_t s) {       // verdict, X <- z, outer test
    const dim3 grid2(EW_BLOCKS, 2);
    PMX_ADA64B(k64b_ada_finish, grid2, b);
    hipLaunchKernelGGL(k64b_ada_decide, dim3(1), dim3(256), 0, s, b);
}
#undef PMX_ADA64B
