// scratch (round 3): how fast can a K1-shaped grid stream Y?  RANDOM data (round 2's ystream.hip ran on zeros), K1's region
// map (256 workgroups x 512 threads, one per CU, 2048 rows x 256 columns per workgroup in 128 x 32 blocks, a barrier per
// block), K1's MFMA load (12 per producer wave + 24 per consumer wave and block = 36 per SIMD) and NOTHING else (no LDS
// operand traffic, no VALU epilogue).  What varies: how Y is fetched, how many blocks are in flight, the row pitch.
//   mode 0  dword loads in the accumulator layout (what k_grad_f16_v8 does: 2 rows x 128 B per wave-instruction)
//   mode 1  dwordx4 loads, 8 rows x 128 B per wave-instruction
//   mode 2  LDS-DMA (global_load_lds_dwordx4) into a ring of DEPTH blocks (4 KB per wave and block), no registers
//   mode 3  no loads
//   mode 4  dwordx4, 64-column blocks (4 rows x 256 B per wave-instruction; 128 x 64 blocks, half as many slots)
// hipcc --offload-arch=gfx950 -O3 -o ystream2 ystream2.hip
#include <hip/hip_runtime.h>
#include <string>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
__device__ __forceinline__ unsigned long long wall_clock64_() { return __builtin_amdgcn_s_memrealtime(); }
#define wall_clock64 wall_clock64_
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_fill(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.37f;
    }
}

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// OPS: where the MFMA operands come from -- 0 two constant registers (the sweep above); 1 one fresh 16-byte fragment per MFMA
// read from LDS, every fragment holding the SAME values (the reads without the switching); 2 the same reads of pseudo-random
// fp16 data (reads + operand switching, as in the real kernel: ~1 fragment per MFMA).  EPI: the producers also run an
// epilogue's worth of work on the Y tile (66 dependent-free VALU, 8 ds_write_b64 per wave and block).
template <int MODE, int DEPTH, bool MFMA, bool NT, int MF = 12, bool BAR = true, int OPS = 0, bool EPI = false>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ Y, int64_t ld, int M, int N, int RP, int gridX, float* out) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    constexpr int BN = MODE == 4 ? 64 : 32, NCB = 256 / BN;
    constexpr int NY = MODE == 4 ? 32 : 16;              // floats per lane and block
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
    if constexpr (OPS != 0) {    // 64 KB of operand fragments at smem + 65536 (the LDS-DMA ring, when used, sits below)
        f16x8* frag = reinterpret_cast<f16x8*>(smem + 65536);
        for (int e = tid; e < 4096; e += 512) {
            f16x8 v;
            for (int i = 0; i < 8; ++i) {
                const unsigned h = (OPS == 2 ? (unsigned)(e * 8 + i) * 2654435761u : (unsigned)((tid & 63) * 8 + i) * 40503u) >> 16;
                v[i] = (_Float16)(((float)(h & 1023) - 512.f) * (OPS == 2 ? 16.f : 0.002f));
            }
            if (OPS == 1) v = (e & 1) ? fb : fa;
            frag[e] = v;
        }
        __syncthreads();
    }
    const f16x8* fragw = reinterpret_cast<const f16x8*>(smem + 65536) + lane;     // + 64 * n: conflict-free 16-byte reads
    const unsigned long long tc0 = __builtin_readcyclecounter(), tw0 = wall_clock64();
    float y[MODE == 2 || MODE == 3 ? 1 : DEPTH][NY] = {};
    const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int slotbuf, int t) {
        const int tc = t < T ? t : T - 1;
        const int rp = tc / NCB, cb = tc % NCB;
        if (!producer) return;
        if constexpr (MODE == 0) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + 4 * hi) * ld + col0 + cb * 32 + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* p = &src[(int64_t)((i & 3) + 8 * (i >> 2)) * ld];
                y[slotbuf][i] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        } else if constexpr (MODE == 1) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + (lane >> 3)) * ld + col0 + cb * 32 + (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4* p = reinterpret_cast<const f32x4*>(src + (int64_t)i * 8 * ld);
                const f32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
                y[slotbuf][4 * i] = v.x; y[slotbuf][4 * i + 1] = v.y; y[slotbuf][4 * i + 2] = v.z; y[slotbuf][4 * i + 3] = v.w;
            }
        } else if constexpr (MODE == 4) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + (lane >> 4)) * ld + col0 + cb * 64 + (lane & 15) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4* p = reinterpret_cast<const f32x4*>(src + (int64_t)i * 4 * ld);
                const f32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
                y[slotbuf][4 * i] = v.x; y[slotbuf][4 * i + 1] = v.y; y[slotbuf][4 * i + 2] = v.z; y[slotbuf][4 * i + 3] = v.w;
            }
        } else if constexpr (MODE == 2) {
            const float* src = Y + (int64_t)(row0 + rp * 128 + w * 32 + (lane >> 3)) * ld + col0 + cb * 32 + (lane & 7) * 4;
            const unsigned dst = ring + (unsigned)(slotbuf * 16384 + w * 4096);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                lds_dma16(src + (int64_t)i * 8 * ld, __builtin_amdgcn_readfirstlane(dst + i * 1024));
        }
    };
    auto use = [&](int slotbuf) {
        if (!producer) return;
        if constexpr (MODE == 2) {
            // the block's tile has landed (vmcnt wait below); read it back like an epilogue would: 4 x 16 B per lane
            const float4* p = reinterpret_cast<const float4*>(smem + slotbuf * 16384 + w * 4096) + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float4 v = p[i * 64]; acc += v.x + v.y + v.z + v.w; }
        } else if constexpr (MODE != 3) {
#pragma unroll
            for (int i = 0; i < NY; ++i) acc += y[slotbuf][i];
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    constexpr int NM = MODE == 4 ? 2 : 1;     // 64-column blocks carry twice the MFMAs
    for (int t = 0; t < T; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if constexpr (MODE == 2) {
                // DEPTH - 1 younger blocks (4 requests each) may stay in flight
                if (producer) {
                    if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
                }
            }
            use(d);
            issue(d, t + d + DEPTH);
            if constexpr (EPI) {
                if (producer) {      // an epilogue's worth: ~4 VALU per value of the tile, two fp16 images written
                    float r[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float x = y[d % (MODE == 2 || MODE == 3 ? 1 : DEPTH)][i & (NY - 1)] * 1.0001f - acc; r[i] = x * 0.5f + (float)(_Float16)x; }
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
            if constexpr (MFMA) {
                const int t0 = (t + d) * 7;
                if (producer) {
                    f16x8 b = fb;
#pragma unroll
                    for (int q = 0; q < MF * NM; ++q) {
                        if constexpr (OPS != 0) { if ((q % 3) != 2) b = fragw[64 * ((t0 + q) & 63)]; }      // 8 of 12: the S fragments
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, b, c0, 0, 0, 0);
                    }
                } else {
                    f16x8 x = fa, z = fb;
#pragma unroll
                    for (int q = 0; q < MF * NM; ++q) {
                        if constexpr (OPS != 0) { if (q & 1) x = fragw[64 * ((t0 + q) & 63)]; else z = fragw[64 * ((t0 + q + 31) & 63)]; }   // one fresh fragment per MFMA pair member
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, z, c0, 0, 0, 0);
                        if constexpr (OPS != 0) { if (q & 1) z = fragw[64 * ((t0 + q + 17) & 63)]; else x = fragw[64 * ((t0 + q + 5) & 63)]; }
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(z, x, c1, 0, 0, 0);
                    }
                }
            }
            if constexpr (BAR) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) out[0] = acc + c0[0] + c1[3];
    if (blockIdx.x == 0 && tid == 0) { out[4] = (float)(__builtin_readcyclecounter() - tc0); out[5] = (float)(wall_clock64() - tw0); }
}

// "hold" mode: ystream2 hold <variant> <seconds> -- one variant back to back for a while (random data, pitch N), so that a
// sampler of the package power / clock (scratch/r3_skeleton_power.py) sees a steady state; prints the average launch time
static int hold(const char* which, double secs) {
    const int M = 16384, N = 16384, RP = 16, gridX = 8;
    const int64_t ld = N;
    float* out; CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(out, 0, 64));
    float* Y; CHECK(hipMalloc(&Y, (size_t)M * ld * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Y, (size_t)M * ld, 1234u); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&](auto kern) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 1024));
        double total_ms = 0; long launches = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(gridX * (N / 256)), dim3(512), 1024, 0, Y, ld, M, N, RP, gridX, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            total_ms += ms; launches += 200;
        }
        printf("hold %s: %.4f ms per launch over %ld launches\n", which, total_ms / launches, launches);
    };
    const std::string w(which);
    const int big = 131072;
    auto go_lds = [&](auto kern) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        double total_ms = 0; long launches = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(gridX * (N / 256)), dim3(512), big, 0, Y, ld, M, N, RP, gridX, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            total_ms += ms; launches += 200;
        }
        printf("hold %s: %.4f ms per launch over %ld launches\n", which, total_ms / launches, launches);
    };
    if (w == "both_v8_ldsconst") { go_lds(k<0, 2, true, true, 12, true, 1, false>); return 0; }     // + operand fragments read from LDS, always the same values
    if (w == "both_v8_ldsrand") { go_lds(k<0, 2, true, true, 12, true, 2, false>); return 0; }       // + pseudo-random operand data
    if (w == "both_v8_ldsrand_epi") { go_lds(k<0, 2, true, true, 12, true, 2, true>); return 0; }    // + an epilogue's VALU and LDS stores
    if (w == "mfma_ldsrand") { go_lds(k<3, 1, true, true, 12, true, 2, false>); return 0; }          // the MFMAs alone, random operands
    if (w == "stream4") go(k<1, 4, false, true>);                 // Y alone, 16-byte requests, four blocks in flight
    else if (w == "stream_v8") go(k<0, 2, false, true>);          // Y alone, v8's request shape
    else if (w == "mfma") go(k<3, 1, true, true>);                // K1's MFMAs alone (36 per SIMD and slot, constant operands)
    else if (w == "both4") go(k<1, 4, true, true>);               // both
    else if (w == "both_v8") go(k<0, 2, true, true>);
    else { printf("unknown variant %s\n", which); return 1; }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && std::string(argv[1]) == "hold") return hold(argv[2], atof(argv[3]));
    const int M = 16384, N = 16384, RP = 16, gridX = 8;
    const int pads[3] = {0, 64, 2048 + 64};
    float* out;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(out, 0, 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pi = 0; pi < 3; ++pi) {
        const int64_t ld = N + pads[pi];
        float* Y;
        CHECK(hipMalloc(&Y, (size_t)M * ld * 4));
        for (int zero = 0; zero < (pi == 0 ? 2 : 1); ++zero) {
            if (zero) CHECK(hipMemset(Y, 0, (size_t)M * ld * 4));
            else { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Y, (size_t)M * ld, 1234u); CHECK(hipDeviceSynchronize()); }
            printf("---- pitch %lld floats (%s data) ----\n", (long long)ld, zero ? "ZERO" : "random");
            auto run = [&](const char* name, auto kern, int lds) {
                CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds > 0 ? lds : 1024));
                float best = 1e9f, tot = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(gridX * (N / 256)), dim3(512), lds > 0 ? lds : 1024, 0, Y, ld, M, N, RP, gridX, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep) { tot += ms / 20; if (ms / 20 < best) best = ms / 20; }
                }
                CHECK(hipGetLastError());
                float h[8]; CHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
                printf("%-44s %.4f ms (best %.4f)  %.0f GB/s   WG0: %.0f cycles in %.1f us = %.0f MHz\n", name, tot / 2, best, (double)M * N * 4 / (tot / 2) / 1e6, h[4], h[5] / 100.0, h[5] > 0 ? h[4] / (h[5] / 100.0) : 0.0);
                fflush(stdout);
            };
            run("no loads, 36 MFMA/SIMD/slot", k<3, 1, true, true>, 0);
            run("no loads, 18 MFMA/SIMD/slot", k<3, 1, true, true, 6>, 0);
            run("no loads, 36 MFMA/SIMD/slot, no barrier", k<3, 1, true, true, 12, false>, 0);
            run("dwordx4 nt d4 + 18 MFMA/SIMD/slot", k<1, 4, true, true, 6>, 0);
            run("dwordx4 nt d4 + 72 MFMA/SIMD/slot", k<1, 4, true, true, 24>, 0);
            run("dwordx4 nt d4 + 36 MFMA, no barrier", k<1, 4, true, true, 12, false>, 0);
            run("dwordx4 nt d4, no MFMA, no barrier", k<1, 4, false, true, 12, false>, 0);
            run("dword nt d2, no MFMA", k<0, 2, false, true>, 0);
            run("dword nt d4, no MFMA", k<0, 4, false, true>, 0);
            run("dword nt d2 + MFMA   (= v8's fetch)", k<0, 2, true, true>, 0);
            run("dword nt d4 + MFMA", k<0, 4, true, true>, 0);
            run("dword plain d2 + MFMA", k<0, 2, true, false>, 0);
            run("dwordx4 nt d2, no MFMA", k<1, 2, false, true>, 0);
            run("dwordx4 nt d4, no MFMA", k<1, 4, false, true>, 0);
            run("dwordx4 nt d2 + MFMA", k<1, 2, true, true>, 0);
            run("dwordx4 nt d4 + MFMA", k<1, 4, true, true>, 0);
            run("dwordx4 nt d8 + MFMA", k<1, 8, true, true>, 0);
            run("LDS-DMA ring 2 + MFMA", k<2, 2, true, true>, 2 * 16384);
            run("LDS-DMA ring 4 + MFMA", k<2, 4, true, true>, 4 * 16384);
            run("LDS-DMA ring 8 + MFMA", k<2, 8, true, true>, 8 * 16384);
            run("LDS-DMA ring 8, no MFMA", k<2, 8, false, true>, 8 * 16384);
            run("dwordx4 64-col blocks d2 + MFMA", k<4, 2, true, true>, 0);
            run("dwordx4 64-col blocks d4 + MFMA", k<4, 4, true, true>, 0);
        }
        CHECK(hipFree(Y));
    }
    return 0;
}
