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
__device__ __forceinline__ double row_sum32_d(double v) {
    v += swz16_d(v);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_XOR1>(v);
    return v;
}
__device__ __forceinline__ double prox64_one(double v, bool ok, const pmx_prox& p, double sk) {
    switch (p.op) {
        case PMX_PROX_ID: return v;
        case PMX_PROX_ZERO: return 0.0;
        case PMX_PROX_PLUS: return v < 0.0 ? 0.0 : v;
        case PMX_PROX_UNITY:
        case PMX_PROX_UNITY_PLUS: {
            if (p.op == PMX_PROX_UNITY_PLUS) v = v < 0.0 ? 0.0 : v;
            const double s = row_sum32_d(ok ? v : 0.0);
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
__device__ __forceinline__ double prox64_row(double v, bool ok, const ProxSeq& ps, double sk) {
    for (int r = 0; r < ps.repeat; ++r)
        for (int q = 0; q < ps.n; ++q) v = prox64_one(v, ok, ps.seq[q], sk);
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
