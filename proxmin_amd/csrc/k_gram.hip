// Lipschitz step rule of the PGM / bSDMM back-ends, on the device.
//
// Restates nmf.step_pgm (W == 1 branch, proxmin/nmf.py:44-65) -> utils.get_spectral_norm
// (proxmin/utils.py:14-35):   step_A = 1 / lmax(S S^T),   step_S = 1 / lmax(A^T A).
// With St = S^T both are 1 / lmax(X^T X) of a tall X (rows x K):
//   k_gram_partial : per-workgroup K x K partial Gram matrices (fp32 products, fp32 accumulation over
//                    at most a few hundred rows)
//   k_gram_reduce  : fixed-order sum of the partials in fp64
//   k_eig          : largest eigenvalue of the symmetric PSD K x K matrix by power iteration
//                    (warm-started from the previous iteration's eigenvector; the reference calls
//                    LAPACK `eigvals` and takes the max), final Rayleigh quotient in fp64.
// No host round trip: the step sizes land in DevStatus::step and are read by the update kernels.
#include "pmx_common.h"

// GRAM_BLOCKS, gram_per, gram_nparts: pmx_common.h (the update kernels that leave partial Gram matrices behind use them too)
constexpr int GRAM_THREADS = 256;
// rows staged per LDS tile: all of a tile's requests go out before the first is used (a dependent round trip to memory
// is ~1.3 us; with 32-row tiles a 128-row share took four of them and the kernel 18 us at 16384 x 64, now 7)
template <int KP> constexpr int gram_chunk() { return KP == 128 ? 64 : 128; }

struct GramArgs {
    const float* X[2];     // factor f: 0 = A (M x K), 1 = St (N x K)
    int64_t rows[2];
    int K;
    float* part;           // [2][GRAM_BLOCKS][KP*KP]
    const DevStatus* status;
    int want[2];           // compute factor f?
};

template <int KP>
__global__ __launch_bounds__(GRAM_THREADS) void k_gram_partial(GramArgs a) {
    constexpr int TS = KP / 16;                 // per-thread micro tile (TS x TS)
    constexpr int GRAM_CHUNK = gram_chunk<KP>(), NLD = GRAM_CHUNK * KP / GRAM_THREADS;
    __shared__ float xs[GRAM_CHUNK][KP + 1];
    if (chain_halted(a.status)) return;
    const int f = blockIdx.y;
    if (!a.want[f]) return;
    const int K = a.K;
    const int64_t rows = a.rows[f];
    const float* X = a.X[f];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    float acc[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int jx = 0; jx < TS; ++jx) acc[i][jx] = 0.f;
    const int64_t per = gram_per(rows);
    const int64_t r0 = (int64_t)blockIdx.x * per;
    if (r0 >= rows) return;                     // (the fold reads gram_nparts(rows) slots)
    const int64_t r1 = r0 + per < rows ? r0 + per : rows;
    for (int64_t rb = r0; rb < r1; rb += GRAM_CHUNK) {
        float ld[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = threadIdx.x + u * GRAM_THREADS;
            const int rr = e / KP, k = e - rr * KP;
            ld[u] = (rb + rr < r1 && k < K) ? X[(rb + rr) * K + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = threadIdx.x + u * GRAM_THREADS;
            xs[e / KP][e % KP] = ld[u];
        }
        __syncthreads();
        const int nrow = r1 - rb < GRAM_CHUNK ? (int)(r1 - rb) : GRAM_CHUNK;     // (rows past the share hold zeros: skipped, not added)
#pragma unroll 4
        for (int rr = 0; rr < nrow; ++rr) {
            float av[TS], bv[TS];
#pragma unroll
            for (int i = 0; i < TS; ++i) av[i] = xs[rr][ti + 16 * i];
#pragma unroll
            for (int i = 0; i < TS; ++i) bv[i] = xs[rr][tj + 16 * i];
#pragma unroll
            for (int i = 0; i < TS; ++i)
#pragma unroll
                for (int jx = 0; jx < TS; ++jx) acc[i][jx] += av[i] * bv[jx];
        }
    }
    float* out = a.part + ((int64_t)f * GRAM_BLOCKS + blockIdx.x) * KP * KP;
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int jx = 0; jx < TS; ++jx) out[(ti + 16 * i) * KP + tj + 16 * jx] = acc[i][jx];
}

struct GramReduceArgs {
    const float* part;     // [2][GRAM_BLOCKS][KP*KP]
    double* G;             // [2][KP*KP]
    int KP;
    const DevStatus* status;
    int want[2];
    int nparts[2];         // valid partial slots of factor f (gram_nparts(rows))
    // [r4] pgm: the stopping test of the iteration whose update kernel ran just before (algorithms.py:130-135) rides in THIS launch
    // as one more workgroup -- it used to be the last-arriving workgroup of k_pgm_update, ~2.5 us behind everybody else; here it runs
    // beside the fold, and k_eig / K1 behind this launch see its verdict (halt) before they touch anything.  nullptr: no test here.
    double* dec_partials;
    DevStatus* dec_status;
    double dec_e_rel[2];
    // bsdmm: the Boyd test of the block whose update kernel ran just before (utils.py:349-391), likewise (dec_bsdmm.status != nullptr)
    BsdmmDecideArgs dec_bsdmm;
};
// Entry e of factor f: EIGHT threads fold up to 32 partials each (two batches of 16 loads in flight: the partials come from other
// XCDs, a load-add-load-add loop pays one memory round trip per term), then the eight sums are added in a fixed order.
// (Round 3: one thread per entry, eight batches: 5.2 us at K = 32; now two round trips.)
__global__ __launch_bounds__(256) void k_gram_reduce(GramReduceArgs a) {
    if (chain_halted(a.status)) return;
    const int n = a.KP * a.KP;
    if ((int)blockIdx.x == (n * 8 + 255) / 256) {          // the extra workgroup: the stopping test (block y == 0 only)
        if (blockIdx.y == 0 && a.dec_partials != nullptr) pgm_decide_body(a.dec_status, a.dec_partials, a.dec_e_rel, 1, false);
        if (blockIdx.y == 0 && a.dec_bsdmm.status != nullptr) {
            __shared__ double dscratch[EW_WAVES];
            bsdmm_decide_body(a.dec_bsdmm, dscratch);
        }
        return;
    }
    const int f = blockIdx.y;
    if (!a.want[f]) return;
    static_assert(GRAM_BLOCKS == 256, "eight threads x 32 partials");
    gram_fold_octet(a.part, a.G, a.KP, a.nparts[f], f, blockIdx.x * 256 + threadIdx.x);      // (pmx_common.h: shared with K1's prologue, k1_gram_fold)
}


struct EigArgs {
    const double* G;       // [2][KP*KP]  (factor f)
    int KP, K;
    DevStatus* status;
    int want[2];
    double scale;          // step = scale / lmax   (user `step = c * step_pgm` support)
    int max_iter;
    double* Q;             // [2][KP*KP] Lanczos basis scratch (exact fallback)
    // small problems (K <= 16, a few thousand rows): the Gram matrix is formed HERE, by this workgroup, instead of by
    // k_gram_partial + k_gram_reduce -- one launch for the whole step rule instead of three (each is pure launch latency
    // at that size).  X[f] == nullptr: G was prepared by the caller.
    const float* X[2];
    int64_t rows[2];
    double* Gw;            // writable view of G
    int force_exact;       // fp64 contexts: always finish with the exact solver (Lanczos + Sturm multisection: lambda to fp64
                           // round-off).  The power iteration runs on the fp32 copy of G and is accepted at a residual of 1e-6
                           // lambda, i.e. lambda to ~1e-12 / (relative gap) -- plenty for fp32 factors, 8e-10 in fp64 ones after
                           // 25 iterations (measured against the reference's fixture)
                           // [r6] 2: the same accuracy from power steps on the fp64 matrix after the fp32 ones (eig_solve_block); the exact
                           // solver only when those do not settle (k_big_f64.hip: K up to 128, where the exact solver costs milliseconds)
};
// one workgroup per factor.  Factor f's eigenvalue sets the step of the OTHER block:
// lmax(A^T A) -> step_S (block 1), lmax(S S^T) -> step_A (block 0)   (nmf.py:44-49)
//
// eig_solve_block: warm-started power iteration + checks + exact fallback, by a whole workgroup of 256 threads.
// G: fp64 Gram matrix (row stride GS; memory or LDS), g: its fp32 copy in LDS [KP][KP+1] (filled by the caller; the first
// barrier inside publishes it).
__device__ __forceinline__ void eig_solve_block(const EigArgs& a, const int f, const double* G, const int GS, float* g) {
    __shared__ double vec[MAXK], wv[MAXK], red[8], eigred2[4];
    DevStatus* st = a.status;
    const int KP = a.KP, K = a.K, ld = KP + 1;
    const int t = threadIdx.x;
    // warm start (all-ones on the first call: the Perron vector of a non-negative Gram matrix is positive)
    {
        double v0 = (t < K) ? st->eigvec[f][t] : 0.0;
        double q = v0 * v0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if ((t & 63) == 0) red[t >> 6] = q;
        __syncthreads();
        const double n0 = sqrt(red[0] + red[1] + red[2] + red[3]);
        if (t < K) vec[t] = (n0 > 0.0 && n0 == n0 && n0 < 1e300) ? v0 / n0 : 1.0 / sqrt((double)K);
    }
    __syncthreads();
    double lam_prev = -1.0, lam = 0.0;
    int it = 0, calm = 0;
    for (; it < a.max_iter; ++it) {
        // w = G v
        if (t < K) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += g[t * ld + k] * (float)vec[k];
            wv[t] = (double)s;
        }
        __syncthreads();
        // norm
        double p = (t < K) ? wv[t] * wv[t] : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
        if ((t & 63) == 0) red[t >> 6] = p;
        __syncthreads();
        const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
        lam = nrm;                         // |G v| with |v| = 1 -> lmax
        __syncthreads();
        if (nrm == 0.0 || !(nrm == nrm)) break;   // zero / NaN matrix
        if (t < K) vec[t] = wv[t] / nrm;
        __syncthreads();
        if (fabs(lam - lam_prev) <= 1e-7 * lam) {
            if (++calm >= 2) { ++it; break; }
        } else calm = 0;
        lam_prev = lam;
    }
    // Rayleigh quotient in fp64 with the fp64 Gram matrix (error quadratic in the eigenvector error)
    if (t < K) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += G[t * GS + k] * vec[k];
        wv[t] = s;
    }
    __syncthreads();
    double num = (t < K) ? wv[t] * vec[t] : 0.0, den = (t < K) ? vec[t] * vec[t] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o); den += __shfl_xor(den, o); }
    __syncthreads();
    if ((t & 63) == 0) { red[t >> 6] = num; red[4 + (t >> 6)] = den; }
    __syncthreads();
    const double rq_n = red[0] + red[1] + red[2] + red[3], rq_d = red[4] + red[5] + red[6] + red[7];
    double l = (rq_d > 0.0) ? rq_n / rq_d : lam;
    if (!(lam == lam)) l = lam;
    // ---- accept only with a small residual |G v - l v| <= 1e-6 l (an eigenvalue lies that close to l);
    //      otherwise (clustered top eigenvalues) fall back to the exact tridiagonal solver below --------
    __syncthreads();
    double rr = (t < K) ? (wv[t] - l * vec[t]) : 0.0;
    rr *= rr;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rr += __shfl_xor(rr, o);
    if ((t & 63) == 0) red[t >> 6] = rr;
    __syncthreads();
    double resid = sqrt((red[0] + red[1] + red[2] + red[3]) / (rq_d > 0.0 ? rq_d : 1.0));
    __syncthreads();
    // ---- [r6] force_exact == 2 (fp64 contexts at size, k_big_f64.hip): the exact solver below is O(K^3) dependent steps through memory
    //      (0.9 ms at K = 64, 3.5 ms at K = 128 -- more than K1).  Instead: power steps ON THE fp64 MATRIX from the converged fp32 vector
    //      until the residual is at 1e-11 l; the Rayleigh quotient's error is quadratic in it, i.e. l is at fp64 round-off.  A dominant
    //      eigenvalue (the Perron root of a non-negative Gram matrix) gets there in a handful of steps; a cluster does not and ends in the
    //      exact solver as before. ----------------------------------------------------------------------------------------------------
    if (a.force_exact == 2 && l > 0.0 && l == l && l < 1e300) {
        for (int r = 0; r < 200 && !(resid <= 1e-11 * l); ++r) {
            double p2 = (t < K) ? wv[t] * wv[t] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) p2 += __shfl_xor(p2, o);
            if ((t & 63) == 0) red[t >> 6] = p2;
            __syncthreads();
            const double nrm2 = sqrt(red[0] + red[1] + red[2] + red[3]);
            __syncthreads();
            if (!(nrm2 > 0.0)) break;
            if (t < K) vec[t] = wv[t] / nrm2;
            __syncthreads();
            if (t < K) {
                double s2 = 0.0;
                for (int k = 0; k < K; ++k) s2 += G[(int64_t)k * GS + t] * vec[k];      // (symmetric: column t, coalesced over t)
                wv[t] = s2;
            }
            __syncthreads();
            double n3 = (t < K) ? wv[t] * vec[t] : 0.0, d3 = (t < K) ? vec[t] * vec[t] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { n3 += __shfl_xor(n3, o); d3 += __shfl_xor(d3, o); }
            if ((t & 63) == 0) { red[t >> 6] = n3; red[4 + (t >> 6)] = d3; }
            __syncthreads();
            const double qn = red[0] + red[1] + red[2] + red[3], qd = red[4] + red[5] + red[6] + red[7];
            __syncthreads();
            l = qd > 0.0 ? qn / qd : l;
            double r3 = (t < K) ? (wv[t] - l * vec[t]) : 0.0;
            r3 *= r3;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r3 += __shfl_xor(r3, o);
            if ((t & 63) == 0) red[t >> 6] = r3;
            __syncthreads();
            resid = sqrt((red[0] + red[1] + red[2] + red[3]) / (qd > 0.0 ? qd : 1.0));
            __syncthreads();
            ++it;
        }
    }
    // ---- a small residual says l is AN eigenvalue, not that it is the largest: the iteration is warm-started from the
    //      previous call's vector, and when two eigenvalues cross (factors with mixed signs under prox_id / soft / hard: no
    //      Perron argument) the iterate can sit on the pair that has just become second.  Two lower bounds on lmax of the
    //      positive semi-definite Gram matrix that such an l would violate: the largest column norm (|G e_i| <= lmax), and
    //      the Rayleigh quotient of three power steps from a fixed vector unrelated to the warm start.  Either above
    //      l (1 + 1e-5) sends the call to the exact solver below. ------------------------------------------------------
    double probe = 0.0;
    {
        // (the fp32 copy of G in LDS is accurate enough for a bound with a 1e-5 margin, and three more trips to the fp64
        // matrix in memory cost 5 us)
        double cn = 0.0;
        if (t < K) {
            for (int k = 0; k < K; ++k) { const double gg = (double)g[k * ld + t]; cn += gg * gg; }
            cn = sqrt(cn);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cn = fmax(cn, __shfl_xor(cn, o));
        if ((t & 63) == 0) red[t >> 6] = cn;
        __syncthreads();
        probe = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        __syncthreads();
        if (t < K) wv[t] = 1.0 + 0.37 * (double)((t * 7) % 5) - 0.61 * (double)(t & 1);      // fixed, sign-mixed start
        __syncthreads();
        double un = 0.0, ud = 1.0;
        for (int stepi = 0; stepi < 3; ++stepi) {
            double s2 = 0.0;
            if (t < K) for (int k = 0; k < K; ++k) s2 += (double)g[k * ld + t] * wv[k];
            double a2 = (t < K) ? s2 * wv[t] : 0.0, b2 = (t < K) ? wv[t] * wv[t] : 0.0, c2 = (t < K) ? s2 * s2 : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a2 += __shfl_xor(a2, o); b2 += __shfl_xor(b2, o); c2 += __shfl_xor(c2, o); }
            __syncthreads();
            if ((t & 63) == 0) { red[t >> 6] = a2; red[4 + (t >> 6)] = b2; eigred2[t >> 6] = c2; }
            __syncthreads();
            un = red[0] + red[1] + red[2] + red[3];
            ud = red[4] + red[5] + red[6] + red[7];
            const double nn = sqrt(eigred2[0] + eigred2[1] + eigred2[2] + eigred2[3]);
            __syncthreads();
            if (t < K) wv[t] = nn > 0.0 ? s2 / nn : 0.0;
            __syncthreads();
        }
        if (ud > 0.0) probe = fmax(probe, un / ud);
    }
    const bool not_dominant = probe > l * (1.0 + 1e-5);
    int used_exact = 0;
    if (l > 0.0 && l == l && l < 1e300 && (!(resid <= (a.force_exact == 2 ? 1e-9 : 1e-6) * l) || not_dominant || a.force_exact == 1)) {
        // ---- Lanczos tridiagonalisation with full re-orthogonalisation (fp64), K steps = exact ---------
        __shared__ double al[MAXK], be[MAXK + 1], cdot[MAXK];
        __shared__ int nT;
        double* Q = a.Q + (int64_t)f * KP * KP;         // Q[j][*] = j-th Lanczos vector
        if (t < K) Q[t] = vec[t];                        // q_0 = current iterate (unit norm)
        if (t == 0) { be[0] = 0.0; nT = K; }
        __syncthreads();
        for (int j = 0; j < K; ++j) {
            __threadfence_block();
            // w = G q_j   (G symmetric: read column t = row t, coalesced over t)
            double w = 0.0;
            if (t < K) {
                const double* qj = Q + (int64_t)j * KP;
                for (int k = 0; k < K; ++k) w += G[(int64_t)k * GS + t] * qj[k];
                wv[t] = w;
            }
            __syncthreads();
            // classical Gram-Schmidt against q_0..q_j, twice ("twice is enough"); alpha_j from the first pass
            for (int pass = 0; pass < 2; ++pass) {
                if (t <= j) {
                    const double* qi = Q + (int64_t)t * KP;
                    double c = 0.0;
                    for (int k = 0; k < K; ++k) c += qi[k] * wv[k];
                    cdot[t] = c;
                }
                __syncthreads();
                if (pass == 0 && t == 0) al[j] = cdot[j];
                if (t < K) {
                    double acc = wv[t];
                    for (int i = 0; i <= j; ++i) acc -= cdot[i] * Q[(int64_t)i * KP + t];
                    wv[t] = acc;
                }
                __syncthreads();
            }
            double q2 = (t < K) ? wv[t] * wv[t] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
            if ((t & 63) == 0) red[t >> 6] = q2;
            __syncthreads();
            const double beta = sqrt(red[0] + red[1] + red[2] + red[3]);
            __syncthreads();
            if (t == 0) be[j + 1] = beta;
            if (j + 1 == K || !(beta > 1e-13 * l)) {      // invariant subspace reached (uniform)
                if (t == 0) nT = j + 1;
                break;
            }
            if (t < K) Q[(int64_t)(j + 1) * KP + t] = wv[t] / beta;
            __syncthreads();
        }
        __syncthreads();
        const int n = nT;
        // ---- largest eigenvalue of T(al, be) by Sturm-count multisection: 256 shifts per round ---------
        __shared__ double blo, bhi;
        __shared__ int firstFull;
        if (t == 0) {
            double lo = 1e300, hi = -1e300;
            for (int i = 0; i < n; ++i) {
                const double rad = (i > 0 ? fabs(be[i]) : 0.0) + (i + 1 < n ? fabs(be[i + 1]) : 0.0);
                lo = fmin(lo, al[i] - rad);
                hi = fmax(hi, al[i] + rad);
            }
            blo = lo; bhi = hi;
        }
        __syncthreads();
        for (int round = 0; round < 9; ++round) {
            const double lo = blo, hi = bhi;
            const double x = lo + (hi - lo) * (double)(t + 1) / 257.0;
            // number of eigenvalues of T smaller than x (LDL^T pivots)
            int cnt = 0;
            double d = 1.0;
            for (int i = 0; i < n; ++i) {
                const double b2 = i > 0 ? be[i] * be[i] : 0.0;
                d = al[i] - x - (i > 0 ? b2 / d : 0.0);
                if (d == 0.0) d = -1e-300;
                cnt += d < 0.0;
            }
            if (t == 0) firstFull = 256;
            __syncthreads();
            if (cnt == n) atomicMin(&firstFull, t);      // smallest shift that has all n eigenvalues below it
            __syncthreads();
            const int ff = firstFull;
            __syncthreads();
            if (t == 0) {
                const double nlo = ff == 0 ? lo : lo + (hi - lo) * (double)ff / 257.0;
                const double nhi = ff == 256 ? hi : lo + (hi - lo) * (double)(ff + 1) / 257.0;
                blo = nlo; bhi = nhi;
            }
            __syncthreads();
        }
        l = 0.5 * (blo + bhi);
        used_exact = 1;
    }
    if (t == 0) {
        st->lam[f] = l;
        st->step[1 - f] = a.scale / l;     // 1/0 -> inf like the reference
        st->eig_iters[f] = used_exact ? -it : it;
    }
    if (t < K) st->eigvec[f][t] = (lam > 0.0 && lam == lam) ? vec[t] : 1.0;
}

// ------------------------------------------------------------------------------------------------
// eig_wave_solve<KM>: eig_solve_block's warm-started power iteration, fp64 Rayleigh quotient, residual and dominance
// checks for K <= KM <= 64, by ONE wave: entry t of a vector in lane t, sums by DPP row_shr shifts inside the rows of 16
// lanes (+ the rows' totals by v_readlane), broadcasts by v_readlane, no barrier anywhere -- the workgroup version spends
// most of its time in ~70 __syncthreads.  Same operations in the same order (the lanes >= K contribute exact zeros).
// Called by the 64 lanes of wave 0 after g[] (fp32 copy of G, zero beyond K) is complete; returns true when the result was
// accepted and stored, false when the exact solver (eig_solve_block from the top) has to be run.
// ------------------------------------------------------------------------------------------------
// [r6] The solve in two halves that do not depend on each other until the verdict: eig_wave_main (warm-started power iteration, fp64 Rayleigh quotient, residual)
// and eig_wave_probe (the dominance probe: largest column norm + three power steps from a fixed sign-mixed vector).  eig_wave_solve runs them one after the other in
// ONE wave (k_eig_small, k_small_front, k64_front: the callers that have one wave to spare); k_eig gives each its own wave (the probe was ~40 % of the solve's time).
// Same operations in the same order either way: bit-identical eigenvalues.
template <int KM>
struct EigWaveOps {
    // sums over the wave by DPP row_shr shifts inside the rows of 16 lanes (+ the rows' totals by v_readlane), broadcasts by v_readlane
    static __device__ __forceinline__ double wsum(double v) {
#define PMX_ROW_SHR(n)                                                                                         \
        {                                                                                              \
            const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + n, 0xf, 0xf, true); \
            const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + n, 0xf, 0xf, true); \
            v += __hiloint2double(hi_, lo_);                                                           \
        }
        PMX_ROW_SHR(1) PMX_ROW_SHR(2) PMX_ROW_SHR(4) PMX_ROW_SHR(8)
#undef PMX_ROW_SHR
        const int hi_ = __double2hiint(v), lo_ = __double2loint(v);
        double tot = __hiloint2double(__builtin_amdgcn_readlane(hi_, 15), __builtin_amdgcn_readlane(lo_, 15));
        if constexpr (KM > 16) {             // the other three DPP rows, in a fixed order
            tot += __hiloint2double(__builtin_amdgcn_readlane(hi_, 31), __builtin_amdgcn_readlane(lo_, 31));
            tot += __hiloint2double(__builtin_amdgcn_readlane(hi_, 47), __builtin_amdgcn_readlane(lo_, 47));
            tot += __hiloint2double(__builtin_amdgcn_readlane(hi_, 63), __builtin_amdgcn_readlane(lo_, 63));
        }
        return tot;
    }
    static __device__ __forceinline__ double bcast(double v, int k) {                   // k uniform
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
    }
};
struct EigWaveMain { double l, lam, vr, resid; int it; };
// t: lane (0 .. 63)
template <int KM>
__device__ __forceinline__ EigWaveMain eig_wave_main(const EigArgs& a, const int t, const double* G, const int GS, const float* g, const int ld, const int K, const double ev0) {
    using O = EigWaveOps<KM>;
    const bool in = t < K;
    const int tt = in ? t : 0;
    const double v0 = in ? ev0 : 0.0;
    const double n0 = sqrt(O::wsum(v0 * v0));
    double vr = in ? ((n0 > 0.0 && n0 == n0 && n0 < 1e300) ? v0 / n0 : 1.0 / sqrt((double)K)) : 0.0;
    double lam_prev = -1.0, lam = 0.0;
    int it = 0, calm = 0;
    for (; it < a.max_iter; ++it) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) sacc += g[tt * ld + k] * (float)O::bcast(vr, k);    // (entries >= K: exact zeros; unrolled so that the LDS reads go out together)
        const double w = in ? (double)sacc : 0.0;
        const double nrm = sqrt(O::wsum(w * w));
        lam = nrm;
        if (nrm == 0.0 || !(nrm == nrm)) break;
        vr = in ? w / nrm : 0.0;
        if (fabs(lam - lam_prev) <= 1e-7 * lam) {
            if (++calm >= 2) { ++it; break; }
        } else calm = 0;
        lam_prev = lam;
    }
    double gv = 0.0;
#pragma unroll
    for (int k = 0; k < KM; ++k) gv += G[tt * GS + k] * O::bcast(vr, k);
    if (!in) gv = 0.0;
    const double rq_n = O::wsum(gv * vr), rq_d = O::wsum(vr * vr);
    double l = (rq_d > 0.0) ? rq_n / rq_d : lam;
    if (!(lam == lam)) l = lam;
    const double r1 = in ? gv - l * vr : 0.0;
    const double resid = sqrt(O::wsum(r1 * r1) / (rq_d > 0.0 ? rq_d : 1.0));
    return EigWaveMain{l, lam, vr, resid, it};
}
template <int KM>
__device__ __forceinline__ double eig_wave_probe(const int t, const float* g, const int ld, const int K) {
    using O = EigWaveOps<KM>;
    const bool in = t < K;
    const int tt = in ? t : 0;
    double cn = 0.0;
    if (in) {
#pragma unroll
        for (int k = 0; k < KM; ++k) { const double gg = (double)g[k * ld + t]; cn += gg * gg; }
        cn = sqrt(cn);
    }
    double probe = 0.0;
#pragma unroll
    for (int k = 0; k < KM; ++k) probe = fmax(probe, O::bcast(cn, k));
    double w2 = in ? 1.0 + 0.37 * (double)((t * 7) % 5) - 0.61 * (double)(t & 1) : 0.0;
    double un = 0.0, ud = 1.0;
    for (int stepi = 0; stepi < 3; ++stepi) {
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < KM; ++k) s2 += (double)g[k * ld + tt] * O::bcast(w2, k);
        if (!in) s2 = 0.0;
        un = O::wsum(s2 * w2);
        ud = O::wsum(w2 * w2);
        const double nn = sqrt(O::wsum(s2 * s2));
        w2 = in ? (nn > 0.0 ? s2 / nn : 0.0) : 0.0;
    }
    if (ud > 0.0) probe = fmax(probe, un / ud);
    return probe;
}
// the verdict, by the lanes of the wave that ran eig_wave_main: accepted -> stored, else the caller runs the exact solver
__device__ __forceinline__ bool eig_wave_verdict(const EigArgs& a, const int f, const int t, const int K, const EigWaveMain& m, const double probe) {
    DevStatus* st = a.status;
    const bool not_dominant = probe > m.l * (1.0 + 1e-5);
    const bool need_exact = m.l > 0.0 && m.l == m.l && m.l < 1e300 && (!(m.resid <= 1e-6 * m.l) || not_dominant || a.force_exact);
    if (!need_exact) {
        if (t == 0) {
            st->lam[f] = m.l;
            st->step[1 - f] = a.scale / m.l;
            st->eig_iters[f] = m.it;
        }
        if (t < K) st->eigvec[f][t] = (m.lam > 0.0 && m.lam == m.lam) ? m.vr : 1.0;
    }
    return !need_exact;
}
template <int KM>
__device__ __forceinline__ bool eig_wave_solve(const EigArgs& a, const int f, const double* G, const int GS, const float* g,
                                               const int ld, const int K, const double ev0) {
    const int t = threadIdx.x;
    const EigWaveMain m = eig_wave_main<KM>(a, t, G, GS, g, ld, K, ev0);
    const double probe = eig_wave_probe<KM>(t, g, ld, K);
    return eig_wave_verdict(a, f, t, K, m, probe);
}

__global__ __launch_bounds__(256) void k_eig(EigArgs a) {
    extern __shared__ __attribute__((aligned(16))) float g[];   // [KP][KP+1]
    if (chain_halted(a.status)) return;
    const int f = blockIdx.x;
    if (!a.want[f]) return;
    const int KP = a.KP, ld = KP + 1;
    const double* G = a.G + (int64_t)f * KP * KP;
    const double ev0 = (int)threadIdx.x < a.K ? a.status->eigvec[f][threadIdx.x] : 0.0;      // (requested with G)
    for (int e = threadIdx.x; e < KP * KP; e += 256) g[(e / KP) * ld + (e % KP)] = (float)G[e];
    if (KP <= 64) {                          // [r6] wave 0 the solve, wave 1 the dominance probe beside it (eig_wave_main / _probe); the exact solver, if needed, is the block version
        __shared__ int s_accepted;
        __shared__ double s_probe;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        EigWaveMain m{};
        if (w == 0) m = KP == 32 ? eig_wave_main<32>(a, lane, G, KP, g, ld, a.K, ev0) : eig_wave_main<64>(a, lane, G, KP, g, ld, a.K, ev0);
        else if (w == 1) {
            const double p = KP == 32 ? eig_wave_probe<32>(lane, g, ld, a.K) : eig_wave_probe<64>(lane, g, ld, a.K);
            if (lane == 0) s_probe = p;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (w == 0) {
            const bool ok = eig_wave_verdict(a, f, lane, a.K, m, s_probe);
            if (lane == 0) s_accepted = ok;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (s_accepted) return;
    }
    eig_solve_block(a, f, G, KP, g);
}

// ------------------------------------------------------------------------------------------------
// k_eig_small<KM>: the whole step rule of a SMALL factor (K <= KM <= 16, a few thousand rows) in one launch and, as far
// as possible, one trip to memory.  At this size nothing is bound by arithmetic or bandwidth: a dependent round trip to
// memory costs ~3000 cycles (1.3 us) and a __syncthreads() that follows a global store waits for its acknowledgement.  So:
//   * everything the kernel will read -- the halt flag, the previous eigenvector, the factor's rows (thread t: rows t,
//     t + 256, ...; four per batch) -- is requested up front, before the first use of any of it;
//   * the Gram matrix is folded in fp64 through LDS and STAYS there; the copy in memory (for pmx_* readers) is written
//     last, and the barriers on the way are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier: no store is waited for);
//   * the solve runs in ONE wave: entry t of a vector in lane t, sums over the 16 lanes of a DPP row by row_shr shifts
//     (~10 cycles a level, a ds_bpermute shuffle > 100), broadcasts by v_readlane; same operations as eig_solve_block,
//     which the rare call that needs the exact solver (clustered or crossing eigenvalues) takes from the top.
// 200 x 1000 x 5: 25 us (Gram prologue inside k_eig, round 2 start) -> 7 us.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
}
constexpr int EIG_SMALL_CH = 32;             // entries of the triangle folded per round
struct EigSmallSmem {
    double Gd[16 * 16];
    float pbuf[EIG_SMALL_CH][256 + 1];
    int s_accepted;
};
// g: [KP][KP+1] floats of (dynamic) shared memory
template <int KM>
__device__ __forceinline__ void eig_small_body(const EigArgs& a, const int f, float* g, EigSmallSmem& sm) {
    constexpr int NP = KM * (KM + 1) / 2;
    constexpr int CH = EIG_SMALL_CH;
    double (&Gd)[16 * 16] = sm.Gd;
    float (&pbuf)[EIG_SMALL_CH][256 + 1] = sm.pbuf;
    int& s_accepted = sm.s_accepted;
    DevStatus* st = a.status;
    if (!a.want[f]) return;
    const int KP = a.KP, K = a.K, ld = KP + 1;
    const int t = threadIdx.x;
    const float* X = a.X[f];
    const int64_t rows = a.rows[f];
    // ---- requests first ------------------------------------------------------------------------------------------
    const int halted = __builtin_nontemporal_load(&st->halt);
    const double ev0 = t < K ? st->eigvec[f][t] : 0.0;
    float acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.f;
    for (int64_t r0 = t; r0 < rows; r0 += 1024) {
        float x[4][KM];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = r0 + 256 * u;
#pragma unroll
            for (int k = 0; k < KM; ++k) x[u][k] = (r < rows && k < K) ? X[r * K + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int idx = 0;
#pragma unroll
            for (int i = 0; i < KM; ++i)
#pragma unroll
                for (int j = i; j < KM; ++j) acc[idx++] += x[u][i] * x[u][j];   // fp32 over a thread's <= 32 rows, fp64 from here on
        }
    }
    if (halted) return;                      // (uniform: one flag)
    // ---- Gram matrix: the threads' partial sums go through LDS, 32 entries of the triangle per round: 16 lanes per entry
    //      add 16 partials each in fp64, a DPP row reduction adds the 16 lanes (fp64 shuffles over 64 lanes cost 8.6 us
    //      here, this 1 us); the matrix stays in LDS -----------------------------------------------------------------------
    for (int e = t; e < KP * ld; e += 256) g[e] = 0.f;
    for (int e = t; e < 16 * 16; e += 256) Gd[e] = 0.0;
#pragma unroll
    for (int c0 = 0; c0 < NP; c0 += CH) {
        if (c0 > 0) lds_barrier();
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c0 + i < NP) pbuf[i][t] = acc[c0 + i];
        lds_barrier();
#pragma unroll
        for (int h = 0; h < CH / 16; ++h) {
            const int e = h * 16 + (t >> 4), part = t & 15;
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += (double)pbuf[e][part + 16 * q];
#define PMX_ROW_SHR(n)                                                                                         \
            {                                                                                                  \
                const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + n, 0xf, 0xf, true);  \
                const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + n, 0xf, 0xf, true);  \
                v += __hiloint2double(hi_, lo_);                                                               \
            }
            PMX_ROW_SHR(1) PMX_ROW_SHR(2) PMX_ROW_SHR(4) PMX_ROW_SHR(8)
#undef PMX_ROW_SHR
            const int idx = c0 + e;
            if (part == 15 && idx < NP) {
                int i = 0, rem = idx;        // idx -> (i, j) of the upper triangle, row-major
                while (rem >= KM - i) { rem -= KM - i; ++i; }
                const int j = i + rem;
                Gd[i * 16 + j] = v;
                Gd[j * 16 + i] = v;
                g[i * ld + j] = (float)v;
                g[j * ld + i] = (float)v;
            }
        }
    }
    lds_barrier();
    const double* G = Gd;
    const int GS = 16;
    if (t < 64) {
        const bool accepted = eig_wave_solve<KM>(a, f, G, GS, g, ld, K, ev0);
        if (t == 0) s_accepted = accepted;
    }
    lds_barrier();
    {   // the copy in memory, for whoever reads G later (row-sharded drivers, tests): nobody in this launch waits for it
        double* Gw = a.Gw + (int64_t)f * KP * KP;
        for (int e = t; e < KP * KP; e += 256) {
            const int i = e / KP, j = e % KP;
            Gw[e] = (i < K && j < K) ? Gd[i * 16 + j] : 0.0;
        }
    }
    if (s_accepted) return;
    eig_solve_block(a, f, G, GS, g);
}
template <int KM>
__global__ __launch_bounds__(256) void k_eig_small(EigArgs a) {
    extern __shared__ __attribute__((aligned(16))) float g[];   // [KP][KP+1]
    __shared__ EigSmallSmem sm;
    eig_small_body<KM>(a, blockIdx.x, g, sm);
}

// ------------------------------------------------------------------------------------------------
// k_small_front<KM>: K1 AND the step rule of a small problem in one launch.  A pgm iteration evaluates the gradient and
// the Lipschitz steps at the same point (algorithms.py:105-106) and the two write disjoint outputs (gradient slabs + loss
// partials; DevStatus::step + the eigenvector), so their workgroups can simply share a grid: tiles of k_grad_small first,
// one workgroup per factor of k_eig_small behind them.  One launch (~3 us) and the longer of two latency chains instead
// of their sum: 200 x 1000 x 5 pgm 26 -> 19 us per iteration.
// ------------------------------------------------------------------------------------------------
template <int KM>
__global__ __launch_bounds__(256) void k_small_front(GradArgs ga, EigArgs ea, int tilesX, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float g[];   // [KP][KP+1] (step-rule role)
    __shared__ union Pool {
        SmallTileSmem<KM> tile;
        EigSmallSmem eig;
        __device__ Pool() {}
    } sm;
    const int b = blockIdx.x;
    if (b < ntiles) grad_small_tile<KM>(ga, sm.tile, b % tilesX, b / tilesX, tilesX);
    else eig_small_body<KM>(ea, b - ntiles, g, sm.eig);
}

void launch_gram(const GramArgs& a, int KP, hipStream_t s) {
    dim3 grid(GRAM_BLOCKS, 2);
    if (KP == 32) hipLaunchKernelGGL(k_gram_partial<32>, grid, dim3(GRAM_THREADS), 0, s, a);
    else if (KP == 64) hipLaunchKernelGGL(k_gram_partial<64>, grid, dim3(GRAM_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_gram_partial<128>, grid, dim3(GRAM_THREADS), 0, s, a);
}
void launch_gram_reduce(const GramReduceArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_gram_reduce, dim3((a.KP * a.KP * 8 + 255) / 256 + 1, 2), dim3(256), 0, s, a);
}
hipError_t launch_eig(const EigArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * a.KP * (a.KP + 1);
    if (a.X[0] != nullptr) {                 // small factors: Gram + solve in one launch
        if (a.K <= 8) hipLaunchKernelGGL(k_eig_small<8>, dim3(2), dim3(256), lds, s, a);
        else hipLaunchKernelGGL(k_eig_small<16>, dim3(2), dim3(256), lds, s, a);
        return hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute((const void*)k_eig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eig, dim3(2), dim3(256), lds, s, a);
    return hipGetLastError();
}
// K1 (k_grad_small's tiles: p.gridY column tiles x p.gridX row tiles) + both step rules in one launch
hipError_t launch_small_front(const GradPlan& p, const GradArgs& ga, const EigArgs& ea, hipStream_t s) {
    const size_t lds = sizeof(float) * ea.KP * (ea.KP + 1);
    const int ntiles = p.gridX * p.gridY;
    if (ga.K <= 8) hipLaunchKernelGGL(k_small_front<8>, dim3(ntiles + 2), dim3(256), lds, s, ga, ea, p.gridY, ntiles);
    else hipLaunchKernelGGL(k_small_front<16>, dim3(ntiles + 2), dim3(256), lds, s, ga, ea, p.gridY, ntiles);
    return hipGetLastError();
}
