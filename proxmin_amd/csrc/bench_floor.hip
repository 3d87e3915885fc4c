// bench_floor.hip -- MEASUREMENT infrastructure of bench.py (not part of the product path, not part of include/pmx.h): what this MI355X package
// can do on K1's own job, measured on the box the bench line is measured on ("the measured (not nominal) peaks ... on the box").
//   pmxf_stream   the SKELETON of K1 for K = 64 (k_grad_f16_v8<HH, RS>): K1's grid and region map (256 workgroups x 512 threads, one per CU,
//                 128 x 32 blocks of a 16384 x 16384 fp32 Y, one barrier per block), K1's MFMA count in today's arithmetic (mode f16x2r:
//                 4 per producer wave + 24 per consumer wave and block = 28 per SIMD and block) and NOTHING ELSE: no gradient is computed.
//                 What varies: how Y is requested (4 / 8 / 16 bytes per lane, or not at all), where the MFMA operands come from (constant
//                 registers, or pseudo-random fp16 fragments read from LDS: the matrix pipe's power depends on operand toggling), and whether
//                 the producers also run an epilogue's worth of VALU work and LDS stores.  Its time is the floor of ANY kernel that streams Y
//                 once and issues these MFMAs on this package under its power cap: K1's distance from it is what is left to win.
//   pmxf_copy     device memcpy (float4 grid-stride kernel), bytes read + written per second
//   pmxf_mfma     dense fp16 MFMA rate (v_mfma_f32_32x32x16_f16, operands in registers, random or zero data), TFLOP/s under the power cap
// Round 3's scratch/ystream2.hip is where this comes from (36 MFMAs per SIMD and block then); kept tracked and built by __graft_entry__.build().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static thread_local char g_err[256] = "";
#define FCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(g_err, sizeof g_err, "%s: %s (line %d)", #x, hipGetErrorString(e_), __LINE__); return -1; } } while (0)

__global__ void kf_fill(float* p, size_t n, unsigned seed, int zero) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = zero ? 0.f : (float)(x >> 8) * (1.0f / 16777216.0f) - 0.37f;
    }
}

// FETCH: 0 = 4 bytes per lane (the accumulator layout: 2 rows x 128 B per wave-instruction), 1 = 8 bytes per lane for a PAIR of blocks (k_grad_f16_v8's
// fetch: adjacent columns belong to the two blocks of a pair, 256 B per row and instruction), 2 = 16 bytes per lane (8 rows x 128 B per instruction),
// 3 = no loads.  Four blocks of Y in flight per producer wave in every mode (64 registers: what K1's producers hold).
// MFP / MFC: MFMAs per block of a producer wave / of a consumer wave (two accumulators: 2 x MFC).  OPS: 0 constant operand registers, 2 one fresh 16-byte
// fragment of pseudo-random fp16 data read from LDS per MFMA.  EPI: the producers also run ~4 VALU per value of the tile and write two fp16 images.
template <int FETCH, int MFP, int MFC, int OPS, bool EPI>
__global__ __launch_bounds__(512, 2) void k_floor(const float* __restrict__ Y, int64_t ld, int M, int N, int RP, int gridX, float* out) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    constexpr int NCB = 8;
    f32x16 c0 = {}, c1 = {};
    f16x8 fa, fb;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (float)((tid * 7 + i * 13) & 255) - 0.1f); fb[i] = (_Float16)(0.002f * (float)((tid * 3 + i * 5) & 127) - 0.11f); }
    const int lin = blockIdx.x, xcd = lin & 7, idx = lin >> 3, gy = N / 256;
    const int rowRegion = idx % gridX, colRegion = xcd * (gy >> 3) + idx / gridX;
    const int row0 = rowRegion * RP * 128, col0 = colRegion * 256;
    const int T = RP * NCB;
    const bool producer = w < 4;
    float acc = out[2];          // opaque: keeps the no-load variants' MFMAs alive
    if constexpr (OPS != 0) {    // 64 KB of operand fragments at smem + 65536
        f16x8* frag = reinterpret_cast<f16x8*>(smem + 65536);
        for (int e = tid; e < 4096; e += 512) {
            f16x8 v;
            for (int i = 0; i < 8; ++i) {
                const unsigned h = ((unsigned)(e * 8 + i) * 2654435761u) >> 16;
                v[i] = (_Float16)(((float)(h & 1023) - 512.f) * 16.f);
            }
            frag[e] = v;
        }
        __syncthreads();
    }
    const f16x8* fragw = reinterpret_cast<const f16x8*>(smem + 65536) + lane;     // + 64 * n: conflict-free 16-byte reads
    float y[FETCH == 3 ? 1 : 4][16] = {};
    auto block_rc = [&](int t, int& rp, int& cb) { const int tc = t < T ? t : T - 1; rp = tc / NCB; cb = tc % NCB; };
    auto issue = [&](int buf, int t) {   // FETCH 0 / 2: block t into buffer buf (= t & 3, a compile-time constant at every call); FETCH 1: the pair (t, t + 1), t even, into buf, buf + 1
        if (!producer) return;
        int rp, cb;
        block_rc(t, rp, cb);
        if constexpr (FETCH == 0) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + 4 * hi) * ld + col0 + cb * 32 + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) y[buf][i] = __builtin_nontemporal_load(&src[(int64_t)((i & 3) + 8 * (i >> 2)) * ld]);
        } else if constexpr (FETCH == 1) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + 4 * hi) * ld + col0 + (cb >> 1) * 64 + 2 * l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(&src[(int64_t)((i & 3) + 8 * (i >> 2)) * ld]));
                y[buf][i] = v[0];
                y[(buf + 1) & 3][i] = v[1];
            }
        } else if constexpr (FETCH == 2) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + (lane >> 3)) * ld + col0 + cb * 32 + (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)i * 8 * ld));
                y[buf][4 * i] = v.x; y[buf][4 * i + 1] = v.y; y[buf][4 * i + 2] = v.z; y[buf][4 * i + 3] = v.w;
            }
        }
    };
    if constexpr (FETCH == 1) { issue(0, 0); issue(2, 2); }
    else if constexpr (FETCH != 3) { issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3); }
    for (int t = 0; t < T; t += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (producer) {
                if constexpr (FETCH != 3) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc += y[d][i];
                }
                if constexpr (EPI) {      // an epilogue's worth: ~4 VALU per value of the tile, two fp16 images written
                    float r[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float x = y[FETCH == 3 ? 0 : d][i] * 1.0001f - acc; r[i] = x * 0.5f + (float)(_Float16)x; }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 h, l;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { h[q] = (_Float16)r[4 * g + q]; l[q] = (_Float16)(r[4 * g + q] - (float)h[q]); }
                        *reinterpret_cast<f16x4*>(smem + 32768 + w * 4096 + g * 1024 + lane * 8) = h;
                        *reinterpret_cast<f16x4*>(smem + 49152 + w * 4096 + g * 1024 + lane * 8) = l;
                    }
                }
            }
            if constexpr (FETCH == 1) { if (d & 1) issue((d + 3) & 3, t + d + 3); }      // the pair (t + d - 1, t + d) is free: blocks t + d + 3, t + d + 4
            else if constexpr (FETCH != 3) issue(d, t + d + 4);
            const int t0 = (t + d) * 7;
            if (producer) {
                f16x8 b = fb;
#pragma unroll
                for (int q = 0; q < MFP; ++q) {
                    if constexpr (OPS != 0) b = fragw[64 * ((t0 + q) & 63)];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, b, c0, 0, 0, 0);
                }
            } else {
                f16x8 x = fa, z = fb;
#pragma unroll
                for (int q = 0; q < MFC; ++q) {
                    if constexpr (OPS != 0) { if (q & 1) x = fragw[64 * ((t0 + q) & 63)]; else z = fragw[64 * ((t0 + q + 31) & 63)]; }   // one fresh fragment per MFMA
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, z, c0, 0, 0, 0);
                    if constexpr (OPS != 0) { if (q & 1) z = fragw[64 * ((t0 + q + 17) & 63)]; else x = fragw[64 * ((t0 + q + 5) & 63)]; }
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(z, x, c1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) out[0] = acc + c0[0] + c1[3];
}

__global__ __launch_bounds__(256) void kf_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// dense fp16 MFMA: 2 waves per SIMD, four independent accumulators per wave, operands in registers (seeded per lane from `seed`: random, or all zero)
__global__ __launch_bounds__(512, 2) void kf_mfma(int iters, unsigned seed, float* out) {
    f16x8 a[2], b[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            unsigned x = (threadIdx.x * 16u + j * 8u + i + blockIdx.x * 8192u) * 2654435761u + seed;
            x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
            a[j][i] = seed ? (_Float16)(((float)(x & 1023) - 512.f) * 0.01f) : (_Float16)0.f;
            b[j][i] = seed ? (_Float16)(((float)((x >> 10) & 1023) - 512.f) * 0.01f) : (_Float16)0.f;
        }
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) s += c[q][0] + c[q][7];
    if (s == 123.456f) out[0] = s;
}

// dense fp64 MFMA (v_mfma_f64_16x16x4_f64): `waves` waves per SIMD, eight independent accumulators per wave, operands in registers -- the roof of k64_grad_pass
typedef double kf_v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void kf_mfma_f64(int iters, unsigned seed, double* out) {
    double a[4], b[2];
    for (int j = 0; j < 4; ++j) {
        unsigned x = (threadIdx.x * 16u + j + blockIdx.x * 8192u) * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
        a[j] = seed ? ((double)(x & 1023) - 512.0) * 0.01 : 0.0;
        if (j < 2) b[j] = seed ? ((double)((x >> 10) & 1023) - 512.0) * 0.01 : 0.0;
    }
    kf_v4d c[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q & 3], b[q >> 2], c[q], 0, 0, 0);
    }
    double s = 0.0;
    for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][3];
    if (s == 123.456) out[0] = s;
}

// [r6] How the matrix pipe takes a slot's worth of DEPENDENT MFMAs (the gSt waves of k_grad_f16_v8<RS> sum a slot in one accumulator):
//   mode 0  24 MFMAs on ONE accumulator, back to back
//   mode 1  the same with the operand reads of the next k step (two 16-byte LDS reads) between every three of them -- the kernel's shape
//   mode 2  as mode 1, the 24 MFMAs alternating between TWO accumulators
//   mode 3  as mode 1, FOUR accumulators (independent: the pipe's own rate with these reads)
// One wave per SIMD and launch (waves = 1) or two (waves = 2: 512 threads); `slots` slots per wave; out: nothing (timing only).
template <int MODE>
__global__ __launch_bounds__(512, 2) void kf_chain(int slots, float* out) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    f16x8* frag = reinterpret_cast<f16x8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 2048; e += blockDim.x) {
        f16x8 v;
        for (int i = 0; i < 8; ++i) v[i] = (_Float16)(((float)((((unsigned)(e * 8 + i) * 2654435761u) >> 16) & 1023) - 512.f) * 0.03f);
        frag[e] = v;
    }
    __syncthreads();
    const f16x8* fw = frag + lane;
    f32x16 c[4] = {};
    f16x8 a0 = fw[0], a1 = fw[64];
    for (int s = 0; s < slots; ++s) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f16x8 r0 = a0, r1 = a1;
            if constexpr (MODE != 0) { r0 = fw[64 * ((2 * ks + s) & 31)]; r1 = fw[64 * ((2 * ks + 1 + s) & 31)]; }
            if constexpr (MODE <= 1) {
                c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, c[0], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, c[0], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, c[0], 0, 0, 0);
            } else if constexpr (MODE == 2) {
                const int e = ks & 1;
                c[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, c[e], 0, 0, 0);
                c[e ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, c[e ^ 1], 0, 0, 0);
                c[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, c[e], 0, 0, 0);
            } else {
                c[(3 * ks) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, c[(3 * ks) & 3], 0, 0, 0);
                c[(3 * ks + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, c[(3 * ks + 1) & 3], 0, 0, 0);
                c[(3 * ks + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, c[(3 * ks + 2) & 3], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
    for (int q = 0; q < 4; ++q) sum += c[q][0] + c[q][9];
    if (sum == 123.456f) out[0] = sum;
}

// cycles-equivalent: average ns per MFMA of one wave (24 per slot), `waves` = 1 or 2 waves per SIMD
extern "C" int pmxf_chain(int device, int mode, int waves, int reps, double* ns_per_mfma) {
    if (!ns_per_mfma || reps <= 0 || (waves != 1 && waves != 2)) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
    FCHECK(hipSetDevice(device));
    float* out = nullptr;
    FCHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    FCHECK(hipEventCreate(&e0));
    FCHECK(hipEventCreate(&e1));
    const int slots = 512;
    double tot = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        FCHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) {
            if (mode == 0) hipLaunchKernelGGL(kf_chain<0>, dim3(256), dim3(256 * waves), 32768, 0, slots, out);
            else if (mode == 1) hipLaunchKernelGGL(kf_chain<1>, dim3(256), dim3(256 * waves), 32768, 0, slots, out);
            else if (mode == 2) hipLaunchKernelGGL(kf_chain<2>, dim3(256), dim3(256 * waves), 32768, 0, slots, out);
            else hipLaunchKernelGGL(kf_chain<3>, dim3(256), dim3(256 * waves), 32768, 0, slots, out);
        }
        FCHECK(hipEventRecord(e1, 0));
        FCHECK(hipEventSynchronize(e1));
        float t = 0.f;
        FCHECK(hipEventElapsedTime(&t, e0, e1));
        if (pass) tot += t / reps;
    }
    FCHECK(hipGetLastError());
    *ns_per_mfma = tot / 2 * 1e6 / ((double)slots * 24.0);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return 0;
}

template <typename Kern>
static int time_launches(Kern kern, int lds, int grid, int reps, const float* Y, int64_t ld, int M, int N, int RP, int gridX, float* out, double* ms) {
    FCHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    FCHECK(hipEventCreate(&e0));
    FCHECK(hipEventCreate(&e1));
    double tot = 0.0;
    for (int pass = 0; pass < 3; ++pass) {       // pass 0 warms the package up
        FCHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, Y, ld, M, N, RP, gridX, out);
        FCHECK(hipEventRecord(e1, 0));
        FCHECK(hipEventSynchronize(e1));
        float t = 0.f;
        FCHECK(hipEventElapsedTime(&t, e0, e1));
        if (pass) tot += t / reps;
    }
    FCHECK(hipGetLastError());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = tot / 2;
    return 0;
}

extern "C" const char* pmxf_last_error() { return g_err; }

// K1's skeleton on a fresh M x N fp32 matrix (M % 2048 == 0, N % 2048 == 0; 16384 x 16384 = cfg3), average ms per launch over 2 x reps launches.
// variant = fetch + 10 * ops + 100 * epi + 1000 * mfma   (fetch 0..3 as FETCH; ops 0 / 2; epi 0 / 1; mfma 0: none, 1: today's 4 + 24, 2: round 3's 12 + 24, 3: K = 128's 8 + 48, 4: a one-gradient pass's 4 + 12)
extern "C" int pmxf_stream(int device, int variant, int M, int N, int zero_data, int reps, double* ms) {
    if (!ms || M <= 0 || N <= 0 || M % 2048 || N % 2048 || reps <= 0) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
    FCHECK(hipSetDevice(device));
    const int fetch = variant % 10, ops = (variant / 10) % 10, epi = (variant / 100) % 10, mf = variant / 1000;
    const int gy = N / 256, gridX = 256 / gy > 0 ? 256 / gy : 1, RP = M / 128 / gridX, grid = gridX * gy;
    float *Y = nullptr, *out = nullptr;
    FCHECK(hipMalloc(&out, 64));
    FCHECK(hipMemset(out, 0, 64));
    FCHECK(hipMalloc(&Y, (size_t)M * N * 4));
    hipLaunchKernelGGL(kf_fill, dim3(4096), dim3(256), 0, 0, Y, (size_t)M * N, 1234u, zero_data);
    FCHECK(hipDeviceSynchronize());
    const int lds = ops || epi ? 131072 : 1024;
    int rc = -2;
#define VAR(F, P, C, O, E) rc = time_launches(k_floor<F, P, C, O, E>, lds, grid, reps, Y, (int64_t)N, M, N, RP, gridX, out, ms)
    const int key = fetch + 10 * (ops ? 1 : 0) + 100 * epi + 1000 * mf;
    switch (key) {
    case 0: VAR(0, 0, 0, 0, false); break;       // the stream alone
    case 1: VAR(1, 0, 0, 0, false); break;
    case 2: VAR(2, 0, 0, 0, false); break;
    case 1003: VAR(3, 4, 12, 0, false); break;   // today's MFMAs alone
    case 1013: VAR(3, 4, 12, 2, false); break;
    case 1000: VAR(0, 4, 12, 0, false); break;   // stream + MFMAs, constant operands
    case 1001: VAR(1, 4, 12, 0, false); break;
    case 1002: VAR(2, 4, 12, 0, false); break;
    case 1010: VAR(0, 4, 12, 2, false); break;   // random operands from LDS
    case 1011: VAR(1, 4, 12, 2, false); break;
    case 1012: VAR(2, 4, 12, 2, false); break;
    case 1111: VAR(1, 4, 12, 2, true); break;    // + an epilogue's VALU and LDS stores
    case 1112: VAR(2, 4, 12, 2, true); break;
    case 3011: VAR(1, 8, 24, 2, false); break;   // [r6] K = 128 (k_grad_f16_k128<HH, RS>): 8 + 48 MFMAs per producer + consumer wave and 128 x 32 block
    case 3013: VAR(3, 8, 24, 2, false); break;   //      ... its MFMAs alone
    case 4011: VAR(1, 4, 6, 2, false); break;    // [r6] a one-gradient pass at K = 64 (bsdmm's K1s): 4 + 12
    case 2001: VAR(1, 12, 12, 0, false); break;  // round 3's 36 MFMAs (mode f16x2), for continuity with profiles/r03_a_ystream2_sweep.txt
    case 2002: VAR(2, 12, 12, 0, false); break;
    default: snprintf(g_err, sizeof g_err, "unknown variant %d", variant); rc = -1;
    }
#undef VAR
    (void)hipFree(Y);
    (void)hipFree(out);
    return rc;
}

extern "C" int pmxf_copy(int device, int64_t bytes, int reps, double* gbs) {
    if (!gbs || bytes < (1 << 20) || reps <= 0) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
    FCHECK(hipSetDevice(device));
    float4 *a = nullptr, *b = nullptr;
    FCHECK(hipMalloc(&a, (size_t)bytes));
    FCHECK(hipMalloc(&b, (size_t)bytes));
    hipLaunchKernelGGL(kf_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float*>(a), (size_t)bytes / 4, 99u, 0);
    hipEvent_t e0, e1;
    FCHECK(hipEventCreate(&e0));
    FCHECK(hipEventCreate(&e1));
    double tot = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        FCHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kf_copy, dim3(256 * 16), dim3(256), 0, 0, a, b, (size_t)bytes / 16);
        FCHECK(hipEventRecord(e1, 0));
        FCHECK(hipEventSynchronize(e1));
        float t = 0.f;
        FCHECK(hipEventElapsedTime(&t, e0, e1));
        if (pass) tot += t / reps;
    }
    FCHECK(hipGetLastError());
    *gbs = 2.0 * (double)bytes / (tot / 2 * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return 0;
}

extern "C" int pmxf_mfma(int device, int random_data, int reps, double* tflops) {
    if (!tflops || reps <= 0) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
    FCHECK(hipSetDevice(device));
    float* out = nullptr;
    FCHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    FCHECK(hipEventCreate(&e0));
    FCHECK(hipEventCreate(&e1));
    const int iters = 2048;      // x 16 MFMAs per wave
    double tot = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        FCHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kf_mfma, dim3(256), dim3(512), 0, 0, iters, random_data ? 4242u : 0u, out);
        FCHECK(hipEventRecord(e1, 0));
        FCHECK(hipEventSynchronize(e1));
        float t = 0.f;
        FCHECK(hipEventElapsedTime(&t, e0, e1));
        if (pass) tot += t / reps;
    }
    FCHECK(hipGetLastError());
    const double flop = 256.0 * 8.0 * (double)iters * 16.0 * 2.0 * 32 * 32 * 16;
    *tflops = flop / (tot / 2 * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return 0;
}

// fp64 matrix-core rate on this box: `waves` (1 or 2) waves per SIMD
extern "C" int pmxf_mfma_f64(int device, int random_data, int waves, int reps, double* tflops) {
    if (!tflops || reps <= 0 || (waves != 1 && waves != 2)) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
    FCHECK(hipSetDevice(device));
    double* out = nullptr;
    FCHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    FCHECK(hipEventCreate(&e0));
    FCHECK(hipEventCreate(&e1));
    const int iters = 4096;      // x 8 MFMAs per wave
    double tot = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        FCHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kf_mfma_f64, dim3(256), dim3(256 * waves), 0, 0, iters, random_data ? 4242u : 0u, out);
        FCHECK(hipEventRecord(e1, 0));
        FCHECK(hipEventSynchronize(e1));
        float t = 0.f;
        FCHECK(hipEventElapsedTime(&t, e0, e1));
        if (pass) tot += t / reps;
    }
    FCHECK(hipGetLastError());
    const double flop = 256.0 * 4.0 * waves * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
    *tflops = flop / (tot / 2 * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return 0;
}
