// fp64 arithmetic for SMALL problems: the pgm / FISTA back-end of nmf() with fp64 operands, products and sums.
// (included by pmx_api.hip after k_grad.hip, k_gram.hip and k_update.hip, whose helpers it uses)
//
// The reference computes in the dtype of its inputs (nmf.py:39-41: `np.dot(A, S) - Y` ...) and every example it ships --
// and BASELINE's cfg1, 200 x 1000 x 5 -- hands it fp64 arrays.  Rounds 1-3 accepted fp64 inputs, computed in fp32 and cast
// back: fine against the north star's rtol 1e-4, six digits short of what an fp64 caller of the reference
// gets.  For the problems the small-problem kernels take (K <= 16, M N <= 2^20, M, N <= 8192: k_grad_small / k_eig_small)
// this file restates the same three launches per iteration in fp64:
//   k64_front<KM>   K1 on 8 x 256 tiles of Y (thread = column, plain fp64 FMAs in a fixed order, gA by 16 lanes per
//                   output + a shuffle tree) and, behind the tiles, one workgroup per factor for the step rule: Gram matrix
//                   with fp64 products (rows staged through LDS 256 at a time, thread = entry (i, j)), then k_eig_small's
//                   own solve -- warm-started power iteration with an fp64 Rayleigh quotient, residual <= 1e-6 lambda,
//                   dominance probes, exact Lanczos / Sturm fall-back (k_gram.hip) -- which was fp64 all along;
//   k64_pgm_update  slab fold, X = prox(Xe - s G), FISTA extrapolation, the two sums of the stopping test
//                   (algorithms.py:93-108,130-135), every operator of proxmin.operators on fp64 values;
//   k_pgm_decide    the stopping test (shared with the fp32 path: it only ever saw fp64 sums).
// Not a fast path (no matrix cores at K <= 16, three launches, ~25 us per iteration at 200 x 1000 x 5): a parity path --
// tests/test_gpu_f64.py holds it to rtol 1e-10 against the reference's own fp64 fixtures.
// ------------------------------------------------------------------------------------------------
constexpr int S64_ROWS = 8;                  // rows of a K1 tile (16 in k_grad_small: the fp64 tile must fit the static LDS limit)

struct Grad64Args {
    const double* Y;
    int64_t ldY;
    const double* A;         // [M][K]
    const double* St;        // [N][K]
    double* slabA;           // [column tiles][M][K]
    double* slabS;           // [row tiles][N][K]
    double* lossPart;        // one per tile
    const DevStatus* status;
    int M, N, K;
    int doA, doS;
};
template <int KM>
struct Tile64Smem {
    double As[S64_ROWS][KM + 1];
    double Ss[KM][SG_COLS + 1];
    double Rs[S64_ROWS][SG_COLS + 1];
    double lred[SG_COLS / 64];
};
// tile (bx, by): rows by * 8 .., columns bx * 256 ..   (grad_small_tile in fp64; nmf.py:13-41 without weights)
template <int KM>
__device__ __forceinline__ void grad64_tile(const Grad64Args& a, Tile64Smem<KM>& sm, int bx, int by, int gx) {
    const int tid = threadIdx.x, K = a.K;
    const int row0 = by * S64_ROWS, col0 = bx * SG_COLS;
    const int n = col0 + tid;
    const bool nval = n < a.N;
    const int halted = __builtin_nontemporal_load(&a.status->halt);
    double sv[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        sv[k] = (nval && k < K) ? a.St[(int64_t)n * K + k] : 0.0;
        sm.Ss[k][tid] = sv[k];
    }
    for (int e = tid; e < S64_ROWS * KM; e += SG_COLS) {
        const int r = e / KM, k = e - r * KM;
        sm.As[r][k] = (row0 + r < a.M && k < K) ? a.A[(int64_t)(row0 + r) * K + k] : 0.0;
    }
    double yv[S64_ROWS];
#pragma unroll
    for (int r = 0; r < S64_ROWS; ++r) yv[r] = (nval && row0 + r < a.M) ? a.Y[(int64_t)(row0 + r) * a.ldY + n] : 0.0;
    if (halted) return;                      // (uniform; nothing has been written yet)
    __syncthreads();
    double gs[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) gs[k] = 0.0;
    double loss = 0.0;
#pragma unroll
    for (int r = 0; r < S64_ROWS; ++r) {
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < KM; ++k) p += sm.As[r][k] * sv[k];
        const double rr = (nval && row0 + r < a.M) ? p - yv[r] : 0.0;
        loss += rr * rr;
        sm.Rs[r][tid] = rr;
#pragma unroll
        for (int k = 0; k < KM; ++k) gs[k] += sm.As[r][k] * rr;
    }
    if (a.doS && nval) {
        double* dst = a.slabS + ((int64_t)by * a.N + n) * K;
        for (int k = 0; k < K; ++k) dst[k] = gs[k];
    }
    __syncthreads();
    if (a.doA & 1) {
        const int sub = tid & 15, og = tid >> 4;
        for (int o0 = 0; o0 < S64_ROWS * K; o0 += SG_COLS / 16) {
            const int o = o0 + og;
            const bool oval = o < S64_ROWS * K;
            const int r = oval ? o / K : 0, k = oval ? o - r * K : 0;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < SG_COLS / 16; ++c) s += sm.Rs[r][sub + 16 * c] * sm.Ss[k][sub + 16 * c];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (oval && sub == 0 && row0 + r < a.M) a.slabA[((int64_t)bx * a.M + row0 + r) * K + k] = s;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
    if ((tid & 63) == 0) sm.lred[tid >> 6] = loss;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < SG_COLS / 64; ++i) s += sm.lred[i];
        a.lossPart[by * gx + bx] = s;
    }
}

// the step rule of factor f (nmf.py:44-65, utils.py:14-35) from fp64 rows: Gram matrix with fp64 products, then the solve
struct Eig64Smem {
    double Gd[16 * 16];
    double chunk[256][16 + 1];               // 256 rows of the factor at a time (odd pitch: entry (i, j) reads columns i and j)
    int s_accepted;
};
template <int KM>
__device__ __forceinline__ void eig64_body(const EigArgs& a, const double* X, const int f, float* g, Eig64Smem& sm) {
    DevStatus* st = a.status;
    if (!a.want[f]) return;
    const int KP = a.KP, K = a.K, ld = KP + 1;
    const int t = threadIdx.x;
    const int64_t rows = a.rows[f];
    const int halted = __builtin_nontemporal_load(&st->halt);
    const double ev0 = t < K ? st->eigvec[f][t] : 0.0;
    if (halted) return;
    const int gi = t >> 4, gj = t & 15;      // this thread's entry of the (padded 16 x 16) Gram matrix
    double acc = 0.0;
    for (int64_t r0 = 0; r0 < rows; r0 += 256) {
        __syncthreads();
        for (int e = t; e < 256 * KM; e += 256) {
            const int rr = e / KM, k = e - rr * KM;
            sm.chunk[rr][k] = (r0 + rr < rows && k < K) ? X[(r0 + rr) * K + k] : 0.0;
        }
        __syncthreads();
        if (gi < KM && gj < KM) {
#pragma unroll 8
            for (int rr = 0; rr < 256; ++rr) acc += sm.chunk[rr][gi] * sm.chunk[rr][gj];   // (x_i x_j == x_j x_i: the matrix is symmetric bit for bit)
        }
    }
    for (int e = t; e < KP * ld; e += 256) g[e] = 0.f;
    sm.Gd[t] = (gi < K && gj < K) ? acc : 0.0;
    __syncthreads();
    if (gi < K && gj < K) g[gi * ld + gj] = (float)acc;
    lds_barrier();
    const double* G = sm.Gd;
    const int GS = 16;
    if (t < 64) {
        const bool accepted = eig_wave_solve<KM>(a, f, G, GS, g, ld, K, ev0);
        if (t == 0) sm.s_accepted = accepted;
    }
    lds_barrier();
    {
        double* Gw = a.Gw + (int64_t)f * KP * KP;
        for (int e = t; e < KP * KP; e += 256) {
            const int i = e / KP, j = e % KP;
            Gw[e] = (i < K && j < K) ? sm.Gd[i * 16 + j] : 0.0;
        }
    }
    if (sm.s_accepted) return;
    eig_solve_block(a, f, G, GS, g);
}
template <int KM>
__global__ __launch_bounds__(256) void k64_front(Grad64Args ga, EigArgs ea, const double* X0, const double* X1, int tilesX, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float g[];   // [KP][KP+1] (step-rule role)
    __shared__ union Pool {
        Tile64Smem<KM> tile;
        Eig64Smem eig;
        __device__ Pool() {}
    } sm;
    const int b = blockIdx.x;
    if (b < ntiles) grad64_tile<KM>(ga, sm.tile, b % tilesX, b / tilesX, tilesX);
    else eig64_body<KM>(ea, b - ntiles ? X1 : X0, b - ntiles, g, sm.eig);
}
// ntiles == 0: the step rule alone (pmx_step_pgm); steps == false: K1 alone (pmx_grad, fixed steps)
hipError_t launch_front64(const Grad64Args& ga, const EigArgs& ea, const double* X0, const double* X1, int tilesX, int tilesY, bool steps, hipStream_t s) {
    const size_t lds = sizeof(float) * ea.KP * (ea.KP + 1);
    const int ntiles = tilesX * tilesY;
    const int nwg = ntiles + (steps ? 2 : 0);
    if (nwg == 0) return hipSuccess;
    if (ga.K <= 8) hipLaunchKernelGGL(k64_front<8>, dim3(nwg), dim3(256), lds, s, ga, ea, X0, X1, tilesX, ntiles);
    else hipLaunchKernelGGL(k64_front<16>, dim3(nwg), dim3(256), lds, s, ga, ea, X0, X1, tilesX, ntiles);
    return hipGetLastError();
}

// every operator of proxmin.operators on ONE fp64 value per lane (K <= 16: a row lives in lanes 0 .. K-1 of its 32)
// -- prox_one (k_update.hip) restated for double, same expressions in the same order (operators.py:20-160)
// sum over the G lanes that share a row (G = 32: a half-wave; 16 / 8: [r4] the single-workgroup adaprox / bsdmm kernels pack 2 / 4 rows
// into a half-wave when K <= 16 / 8).  Lanes >= K hold zeros, so the 32-lane butterfly and the shorter ones give the same bits.
template <int G>
__device__ __forceinline__ double row_sum_d(double v) {
    if (G > 16) v += swz16_d(v);
    if (G > 8) v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_XOR1>(v);
    return v;
}
template <int G = 32>
__device__ __forceinline__ double prox64_one(double v, bool ok, const pmx_prox& p, double sk) {
    switch (p.op) {
        case PMX_PROX_ID: return v;
        case PMX_PROX_ZERO: return 0.0;
        case PMX_PROX_PLUS: return v < 0.0 ? 0.0 : v;
        case PMX_PROX_UNITY:
        case PMX_PROX_UNITY_PLUS: {
            if (p.op == PMX_PROX_UNITY_PLUS) v = v < 0.0 ? 0.0 : v;
            const double s = row_sum_d<G>(ok ? v : 0.0);
            return v / s;                                   // (no zero guard, like the reference)
        }
        default: {
            const double t = p.relative ? p.thresh * sk : p.thresh;
            double x = v;
            switch (p.op) {
                case PMX_PROX_MIN: x = (x - t < 0.0) ? t : x; break;
                case PMX_PROX_MAX: x = (x - t > 0.0) ? t : x; break;
                case PMX_PROX_HARD: x = (fabs(x) < t) ? 0.0 : x; break;
                case PMX_PROX_HARD_PLUS: x = (fabs(x) < t) ? 0.0 : x; x = x < 0.0 ? 0.0 : x; break;
                case PMX_PROX_SOFT:
                case PMX_PROX_SOFT_PLUS: {
                    double m = fabs(x) - t;
                    m = m < 0.0 ? 0.0 : m;
                    const double sg = (double)(x > 0.0) - (double)(x < 0.0);
                    x = sg * m;
                    if (p.op == PMX_PROX_SOFT_PLUS) x = x < 0.0 ? 0.0 : x;
                    break;
                }
                default: break;
            }
            return x;
        }
    }
}
template <int G = 32>
__device__ __forceinline__ double prox64_row(double v, bool ok, const ProxSeq& ps, double sk) {
    for (int r = 0; r < ps.repeat; ++r)
        for (int q = 0; q < ps.n; ++q) v = prox64_one<G>(v, ok, ps.seq[q], sk);
    return v;
}

struct Pgm64Args {
    double* X[2];
    double* Xe[2];           // extrapolated point (== X when not accelerated)
    double* G[2];            // gradient output (pgm's second return value)
    const double* slab[2];
    int nslab[2];
    int64_t rows[2];
    int K;
    ProxSeq prox[2];
    DevStatus* status;
    double* partials;
    int accelerated;
    double omega_next;
};
// grid (workgroups, 2 blocks); half-wave = row as in k_pgm_update; idle workgroups of the partial-sum slots stay zero
__global__ __launch_bounds__(EW_THREADS) void k64_pgm_update(Pgm64Args a) {
    __shared__ double scratch[2 * EW_WAVES];
    const int j = blockIdx.y;
    const int halted = __builtin_nontemporal_load(&a.status->halt);
    const double s = a.status->step[j];
    if (halted) return;
    const int64_t rows = a.rows[j];
    const int K = a.K;
    const int l32 = threadIdx.x & 31;
    const bool ok = l32 < K;
    double d2 = 0.0, n2 = 0.0;
    const int64_t hw = ((int64_t)blockIdx.x * EW_THREADS + threadIdx.x) >> 5, nhw = ((int64_t)gridDim.x * EW_THREADS) >> 5;
    for (int64_t r = hw; r < rows; r += nhw) {
        const int64_t e = r * K + l32;
        double g = 0.0;
        if (ok)
            for (int q = 0; q < a.nslab[j]; ++q) g += a.slab[j][(int64_t)q * rows * K + e];      // fixed order: slab 0, 1, 2, ...
        const double xo = ok ? a.X[j][e] : 0.0;
        const double xe = a.accelerated ? (ok ? a.Xe[j][e] : 0.0) : xo;
        double v = xe - s * g;                                               // algorithms.py:107-108
        v = prox64_row(v, ok, a.prox[j], s);
        if (ok) {
            a.X[j][e] = v;
            a.G[j][e] = g;
            if (a.accelerated) a.Xe[j][e] = v + a.omega_next * (v - xo);     // algorithms.py:93-95 of the next iteration
            const double d = v - xo;
            d2 += d * d;
            n2 += v * v;
        }
    }
    double red[2] = {d2, n2};
    block_sum_store<2>(red, part_ptr(a.partials, SL_DIFF2, j) + blockIdx.x, (int64_t)2 * EW_BLOCKS, scratch);
}
void launch_pgm64_update(const Pgm64Args& a, int nbx, hipStream_t s) { hipLaunchKernelGGL(k64_pgm_update, dim3(nbx, 2), dim3(EW_THREADS), 0, s, a); }

// fold of the gradient slabs alone (pmx_grad in an fp64 context)
struct Fold64Args {
    const double* slab[2];
    int nslab[2];
    double* G[2];
    int64_t count[2];
};
__global__ __launch_bounds__(256) void k64_fold(Fold64Args a) {
    const int j = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.count[j]; e += (int64_t)gridDim.x * 256) {
        double g = 0.0;
        for (int q = 0; q < a.nslab[j]; ++q) g += a.slab[j][(int64_t)q * a.count[j] + e];
        a.G[j][e] = g;
    }
}
void launch_fold64(const Fold64Args& a, hipStream_t s) { hipLaunchKernelGGL(k64_fold, dim3(64, 2), dim3(256), 0, s, a); }

// ------------------------------------------------------------------------------------------------
// [r4] adaprox in fp64 (small problems): the whole tail of an iteration (algorithms.py:369-410, nmf.py:91-93) by ONE workgroup.
// Half-wave = row (K <= 16 lanes of its 32 active), the 32 half-waves of the workgroup stride over the rows; every sum over a
// block is a wave tree + a 16-term serial sum in LDS, the same for every thread.  State (X, X_, M, V, Vhat, Psi, z) in global
// memory -- <= 1 MB per array, L2-resident.  Not a fast path (a proximal pass over 8192 rows is 256 trips of the loop): the parity
// path for fp64 callers of adaprox, as k64_pgm_update is for pgm.
// ------------------------------------------------------------------------------------------------
struct Ada64Args {
    double* X[2];
    double* Xp[2];           // pre-update iterate (check_convergence)
    double* Mm[2];
    double* Vv[2];
    double* Vh[2];           // nullptr unless warm-started (algorithms.py:356-359)
    double* Psi[2];
    double* z[2];
    const double* slab[2];
    int nslab[2];
    int64_t rows[2];
    int K;
    ProxSeq prox[2];
    int has_prox[2];
    DevStatus* status;
    int scheme, it;
    double b1t, b1prev, b2, eps, p;
    int check_convergence;
    double e_rel[2];
    int prox_max_iter;
    int use_fixed;
    double fixed[2];
    double* alpha_out;       // [2][16]: the steps this iteration used (pmx_step_adaprox in an fp64 context reads them)
};
template <int NV>
__device__ __forceinline__ void wg_sum(double (&v)[NV], double* sm /* >= NV * EW_WAVES */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) sm[i * EW_WAVES + w] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int q = 0; q < EW_WAVES; ++q) s += sm[i * EW_WAVES + q];
        v[i] = s;
    }
}
// G lanes per row (8 / 16 / 32 for K <= 8 / 16 / ..): 1024 / G rows per trip of the loops
template <int G>
__global__ __launch_bounds__(EW_THREADS) void k64_ada_iter(Ada64Args a) {
    constexpr int NG = EW_THREADS / G;       // row groups of the workgroup
    __shared__ double sm[2 * EW_WAVES];
    __shared__ double cs[NG][17];
    __shared__ double alpha_s[2][16];
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    const int tid = threadIdx.x, l32 = tid % G, hw = tid / G;
    const int K = a.K;
    const bool ok = l32 < K;
    // ---- step sizes from the CURRENT factors, both blocks (Jacobi: algorithms.py:370 -> nmf.py:93: mean over the rows / 10)
    for (int j = 0; j < 2; ++j) {
        double s = 0.0;
        if (ok)
            for (int64_t r = hw; r < a.rows[j]; r += NG) s += a.X[j][r * K + l32];
        __syncthreads();
        if (ok) cs[hw][l32] = s;
        __syncthreads();
        if (tid < 16) {
            double t = 0.0;
            if (tid < K)
                for (int q = 0; q < NG; ++q) t += cs[q][tid];
            const double al = tid >= K ? 0.0 : (a.use_fixed ? a.fixed[j] : (t / (double)a.rows[j]) / 10.0);
            alpha_s[j][tid] = al;
            a.alpha_out[j * 16 + tid] = al;
        }
    }
    __syncthreads();
    // ---- scalars of the moment schemes (algorithms.py:147-245), all in fp64
    const double b1 = a.b1t, b2 = a.b2, t = (double)(a.it + 1);
    const double bias1 = 1.0 - pow(b1, t), bias2 = 1.0 - pow(b2, t);
    const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
    const double rho = rho_inf - 2.0 * t * pow(b2, t) / (1.0 - pow(b2, t));
    const double rfac = rho > 4.0 ? sqrt((rho - 4.0) * (rho - 2.0) * rho_inf / (rho_inf - 4.0) / (rho_inf - 2.0) / rho) : 1.0;
    const double xfac = ((1.0 - b1) * (1.0 - b1)) / ((1.0 - a.b1prev) * (1.0 - a.b1prev));
    int taus[2] = {0, 0};
    for (int j = 0; j < 2; ++j) {
        const int64_t rows = a.rows[j];
        const double alpha = ok ? alpha_s[j][l32] : 0.0;
        double* X = a.X[j];
        // ---- moments and update (algorithms.py:375-378)
        double maxpsi = -1.0;
        for (int64_t r = hw; r < rows; r += NG) {
            if (!ok) continue;
            const int64_t e = r * K + l32;
            double g = 0.0;
            for (int q = 0; q < a.nslab[j]; ++q) g += a.slab[j][(int64_t)q * rows * K + e];      // fixed order: slab 0, 1, 2, ...
            const double m = (1.0 - b1) * g + b1 * a.Mm[j][e];
            const double v = (1.0 - b2) * (g * g) + b2 * a.Vv[j][e];
            a.Mm[j][e] = m;
            a.Vv[j][e] = v;
            double phi, psi;
            switch (a.scheme) {
                case PMX_ADAM: phi = m / bias1; psi = sqrt(v / bias2) + a.eps; break;
                case PMX_NADAM: phi = (b1 * m + (1.0 - b1) * g) / bias1; psi = sqrt(v / bias2) + a.eps; break;
                case PMX_RADAM:
                    phi = m / bias1;
                    psi = rho > 4.0 ? sqrt(v / bias2) / rfac : 1.0;
                    if (a.eps > 0.0) psi = fmax(psi, sqrt(a.eps));
                    break;
                default: {   // amsgrad / padam / adamx (algorithms.py:170-221)
                    double cap = v;
                    if (a.Vh[j] != nullptr) {
                        const double old = a.Vh[j][e];
                        cap = fmax(a.scheme == PMX_ADAMX ? xfac * old : old, v);
                        a.Vh[j][e] = cap;
                    }
                    if (a.eps > 0.0) cap = fmax(cap, a.eps);
                    psi = a.scheme == PMX_PADAM ? pow(cap, a.p) : sqrt(cap);
                    phi = m;
                }
            }
            const double xo = X[e];
            if (a.check_convergence) a.Xp[j][e] = xo;
            X[e] = xo - alpha * phi / psi;
            if (a.has_prox[j]) { a.Psi[j][e] = psi; a.z[j][e] = X[e]; }
            maxpsi = nanmax(maxpsi, psi);
        }
        // ---- the proximal sub-iterations (algorithms.py:380-400)
        if (a.has_prox[j]) {
            double mv = wave_nanmax(maxpsi);
            __syncthreads();
            if ((tid & 63) == 0) sm[tid >> 6] = mv;
            __syncthreads();
            double mp = sm[0];
            for (int q = 1; q < EW_WAVES; ++q) mp = nanmax(mp, sm[q]);
            const double gamma = alpha / mp;                  // :384
            const double rat = gamma / alpha;                 // NaN if alpha == 0, as in the reference
            int tau = 0;
            for (tau = 1; tau <= a.prox_max_iter; ++tau) {
                double red[2] = {0.0, 0.0};
                for (int64_t r = hw; r < rows; r += NG) {     // (whole half-waves take or skip a row: the row sums of prox_unity* need all lanes)
                    const int64_t e = r * K + l32;
                    const double zz = ok ? a.z[j][e] : 0.0, x = ok ? X[e] : 0.0, ps = ok ? a.Psi[j][e] : 0.0;
                    double v = zz - rat * ps * (zz - x);
                    v = prox64_row<G>(v, ok, a.prox[j], gamma);
                    if (ok) {
                        const double d = v - zz;
                        red[0] += d * d;
                        red[1] += zz * zz;
                        a.z[j][e] = v;
                    }
                }
                wg_sum<2>(red, sm);
                if (red[0] <= a.e_rel[j] * a.e_rel[j] * red[1]) break;
            }
            if (tau > a.prox_max_iter) tau = a.prox_max_iter;
            taus[j] = tau;
            for (int64_t r = hw; r < rows; r += NG)
                if (ok) X[r * K + l32] = a.z[j][r * K + l32];      // X[j][:] = z (:400)
        }
    }
    // ---- outer stopping test (algorithms.py:403-410) and bookkeeping
    int conv[2] = {0, 0};
    double nrm[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    if (a.check_convergence) {
        for (int j = 0; j < 2; ++j) {
            double red[2] = {0.0, 0.0};
            if (ok)
                for (int64_t r = hw; r < a.rows[j]; r += NG) {
                    const int64_t e = r * K + l32;
                    const double x = a.X[j][e], d = x - a.Xp[j][e];
                    red[0] += d * d;
                    red[1] += x * x;
                }
            wg_sum<2>(red, sm);
            nrm[j][0] = red[0]; nrm[j][1] = red[1];
            conv[j] = red[0] <= a.e_rel[j] * a.e_rel[j] * red[1];
        }
    }
    if (tid == 0) {
        for (int j = 0; j < 2; ++j) {
            st->last_tau[j] = taus[j];
            st->sub_tau[j] = taus[j];
            st->sub_total[j] += taus[j];
            st->conv[j] = conv[j];
            st->norms[j][0] = nrm[j][0]; st->norms[j][1] = nrm[j][1];
        }
        st->it_done += 1;
        if (a.check_convergence && conv[0] && conv[1]) {
            st->stopped = 1;
            st->reason = HALT_CONVERGED;
            __threadfence();
            st->halt = 1;
        }
    }
}
void launch_ada64_iter(const Ada64Args& a, hipStream_t s) {
    if (a.K <= 8) hipLaunchKernelGGL(k64_ada_iter<8>, dim3(1), dim3(EW_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k64_ada_iter<16>, dim3(1), dim3(EW_THREADS), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// [r4] bSDMM in fp64 (small problems): one block update -- X_j, the constraint variables Z_i / U_i, the residual norms and
// the convergence test of the block (algorithms.py:805-844, utils.py:269-391; identity L, steps_g_update = "steps_f") -- by ONE
// workgroup, after k64_front left the block's gradient slabs and step (nmf.py:181-193).  k_bsdmm_update / k_bsdmm_decide in fp64.
// ------------------------------------------------------------------------------------------------
struct Bsdmm64Args {
    double* X;
    const double* slab;
    int nslab;
    double* Z[PMX_MAX_G];
    double* U[PMX_MAX_G];
    int64_t rows;
    int K, j, n_g;
    ProxSeq prox_f;
    ProxSeq prox_g[PMX_MAX_G];
    DevStatus* status;
    double e_rel, e_abs;
    int last_block;
};
template <int G>
__global__ __launch_bounds__(EW_THREADS) void k64_bsdmm_block(Bsdmm64Args a) {
    constexpr int NG = EW_THREADS / G;
    __shared__ double sm[(2 + 4 * PMX_MAX_G) * EW_WAVES];
    DevStatus* st = a.status;
    if (chain_halted(st)) return;
    const int tid = threadIdx.x, l32 = tid % G, hw = tid / G;
    const int K = a.K, j = a.j;
    const bool ok = l32 < K;
    const double sf = st->step[j];
    const double sg = sf * 1.0 * 2.0 * (double)a.n_g;          // get_step_g (utils.py:269-279), identity L
    const double w = a.n_g > 0 ? sf / sg : 0.0;                // step_f / step_g[i]
    const double nisg = a.n_g > 0 ? -1.0 / sg : 0.0;
    double red[2 + 4 * PMX_MAX_G];
#pragma unroll
    for (int i = 0; i < 2 + 4 * PMX_MAX_G; ++i) red[i] = 0.0;
    for (int64_t r = hw; r < a.rows; r += NG) {                // (whole half-waves take or skip a row: prox_unity* sums over its lanes)
        const int64_t e = r * K + l32;
        double g = 0.0;
        if (ok)
            for (int q = 0; q < a.nslab; ++q) g += a.slab[(int64_t)q * a.rows * K + e];
        const double xo = ok ? a.X[e] : 0.0;
        double dx = 0.0;
        for (int i = 0; i < a.n_g; ++i)                         // utils.py:330-336
            if (ok) dx += w * (xo - a.Z[i][e] + a.U[i][e]);
        double v = (xo - dx) - sf * g;                          // utils.py:338 + nmf.py:185
        v = prox64_row<G>(v, ok, a.prox_f, sf);
        if (ok) {
            a.X[e] = v;
            const double d = v - xo;
            red[0] += d * d;
            red[1] += v * v;
        }
        for (int i = 0; i < a.n_g; ++i) {                       // do_the_mm, utils.py:295-304
            const double zo = ok ? a.Z[i][e] : 0.0, uo = ok ? a.U[i][e] : 0.0;
            double zn = v + uo;
            zn = prox64_row<G>(zn, ok, a.prox_g[i], sg);
            if (ok) {
                const double rr = v - zn, sd = nisg * (zn - zo), un = uo + rr, us = un / sg;
                a.Z[i][e] = zn;
                a.U[i][e] = un;
                red[2 + 4 * i + 0] += rr * rr;
                red[2 + 4 * i + 1] += sd * sd;
                red[2 + 4 * i + 2] += zn * zn;
                red[2 + 4 * i + 3] += us * us;
            }
        }
    }
    wg_sum<2 + 4 * PMX_MAX_G>(red, sm);
    if (tid == 0) {     // check_constraint_convergence (utils.py:349-391) and end-of-iteration bookkeeping
        const double sq = sqrt((double)(a.rows * K));
        int conv = 1;
        if (a.n_g == 0) {
            const double e_pri = sq * a.e_abs + a.e_rel * sqrt(red[1]);
            const double e_dual = sq * a.e_abs + a.e_rel * 0.0;
            conv = (0.0 <= e_pri) && (sqrt(red[0]) <= e_dual);
        } else {
            for (int i = 0; i < a.n_g; ++i) {
                const double lR = sqrt(red[2 + 4 * i + 0]), lS = sqrt(red[2 + 4 * i + 1]);
                const double e_pri = sq * a.e_abs + a.e_rel * fmax(sqrt(red[1]), sqrt(red[2 + 4 * i + 2]));
                const double e_dual = sq * a.e_abs + a.e_rel * sqrt(red[2 + 4 * i + 3]);
                conv &= (lR <= e_pri) && (lS <= e_dual);
            }
        }
        st->conv[j] = conv;
        st->norms[j][0] = red[0];
        st->norms[j][1] = red[1];
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
void launch_bsdmm64_block(const Bsdmm64Args& a, hipStream_t s) {
    if (a.K <= 8) hipLaunchKernelGGL(k64_bsdmm_block<8>, dim3(1), dim3(EW_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k64_bsdmm_block<16>, dim3(1), dim3(EW_THREADS), 0, s, a);
}
