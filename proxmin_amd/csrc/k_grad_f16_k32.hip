// K1 for K = 32 in the two-term fp16 arithmetic of k_grad_f16_v8 (included by pmx_api.hip after k_grad_bf16.hip, whose image
// layouts, swizzles, transposing reads, operand scales and GradV4Args it shares): BASELINE cfg2's shape (4096 x 4096, K = 32)
// in mode f16x2, which rounds 1-3 ran on the split-bf16 fall-backs (10 k it/s against 13 k in exact fp32).
//
// Same pass, same roles, same images as k_grad_f16_v8 -- four producer waves (P = A S, R = P - Y, R split into two fp16 terms and
// parked as [n][m] images), four consumer waves two blocks behind (gA = R S^T, gSt = R^T A), one barrier per 128 x 32 block, the
// region's S terms resident in LDS, Y straight into registers in column-interleaved block pairs -- with half the contraction
// length:
//   * the images keep v8's 128-byte rows (64 k slots) and use the first four 16-byte chunks of a row: every swizzle, b128 read
//     and transposing read of v8 applies unchanged to the k steps that exist (ks = 0, 1); the other half of a row is never read;
//   * P: 6 MFMAs per block (2 k steps x 3 products) instead of 12; gA: one 32-wide k tile per wave (6 MFMAs); gSt: there is only
//     one k tile per block, so the consumer waves split the region's BLOCKS instead of the panel's rows: wave j contracts blocks
//     2 j and 2 j + 1 over all 128 rows (24 MFMAs in those two slots of a panel, none in the other six -- a slot is far from
//     matrix-pipe bound here) and holds two accumulator tiles instead of eight quarter tiles: ONE gSt slab per row region, no
//     partial sums across waves (a first version split the rows in quarters: four slabs per row region, 32 MB more for the
//     update kernel to fold at cfg2);
//   * the slot keeps v8's hand-laid order: twelve steps, the MFMAs in the first six, the epilogue of block s - 1 in the first
//     eight, the sixteen Y requests of the next pair in the last four.
// No chains (16 column regions at 4096 columns: 16 gA slabs of 512 KB), no weights (a weighted context at this shape runs the
// split-bf16 kernels or reopens in fp32: engine.open_weighted).  LDS as v8: 128 KB.
// ------------------------------------------------------------------------------------------------
// R3 [r4]: mode f16x2r -- the third fp16 terms of A and S in the residual's product and a second accumulator, as in k_grad_f16_v8<.., R3>
// (why: that kernel's header): 5 instead of 3 MFMAs per k step in the producers, S's third term in LDS (160 KB in all).
template <bool LOSS, bool R3 = false>
__global__ __launch_bounds__(V5_THREADS, 2) void k_grad_f16_k32(GradV4Args a) {
    constexpr int K = 32, ROWB = 128, NCB = V5_NB, KS = 2;
    constexpr int NT = R3 ? 3 : 2;
    constexpr int SLB = NT * V5_S_TERM, OFF_A = NCB * SLB, OFF_R = OFF_A + V5_AIMG_BYTES;
    static_assert(OFF_R + 2 * V5_R_BYTES <= 160 * 1024 && OFF_R % 256 == 0, "");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    if (chain_halted(a.status)) return;
    k1_gram_fold(a.fold);                    // [r6] the step rule's Gram fold of THIS iteration, in the first workgroups (pmx_common.h)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int li = lane & 15, lq = lane >> 4;
    const int M = a.M, N = a.N;
    int rowRegion, colRegion;
    {
        const int lin = blockIdx.x, gx = a.gridX, gy = a.gridY;
        if (gy % 8 == 0) {                   // consecutive workgroups land on consecutive XCDs: an XCD takes a band of column regions
            const int xcd = lin & 7, idx = lin >> 3;
            rowRegion = idx % gx;
            colRegion = xcd * (gy >> 3) + idx / gx;
        } else {
            rowRegion = lin % gx;
            colRegion = lin / gx;
        }
    }
    const int row0 = rowRegion * a.RP * V5_BM;
    const int col0 = colRegion * NCB * V5_BN;          // N % 256 == 0: every region has all 8 column blocks
    // v8's column map: blocks 2q and 2q + 1 share the 64 columns col0 + 64 q .. (even columns to the even block)
    auto block_col = [&](int b, int n) { return col0 + 64 * (b >> 1) + 2 * n + (b & 1); };
    int nrp = (M - row0 + V5_BM - 1) / V5_BM;
    if (nrp > a.RP) nrp = a.RP;
    if (nrp < 0) nrp = 0;
    const int T = nrp * NCB;                 // blocks of this region (even); slots = T + 2
    const bool producer = w < 4;
    const int j = w & 3;                     // index within the role
    float lossAcc = 0.f;

    if (T <= 0) {                              // region outside the matrix: its gSt slab parts and loss partial are zero
        if (!producer) {
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
            for (int c = 2 * j; c < 2 * j + 2; ++c)
                for (int i = 0; i < 16; ++i) {
                    const int gn = block_col(c, tile_row(i, lane));
                    if (gn < N && a.doS) dst[(int64_t)gn * K + l31] = 0.f;
                }
        }
        if (tid == 0) a.lossPart[blockIdx.x] = 0.0;
        return;
    }

    // [r5] the region's S rows are REQUESTED first: they do not depend on the scales, and the launch opened with three dependent round trips to
    // memory (factor maxima -> S rows -> A rows and the first Y tiles), ~1.5 us each -- a tenth of this kernel at cfg2's size
    const int row64 = tid >> 3, f4 = tid & 7, r = row64 & 31;
    float4 sr[NCB / 2];
#pragma unroll
    for (int c2 = 0; c2 < NCB / 2; ++c2)
        sr[c2] = reinterpret_cast<const float4*>(a.St + (int64_t)block_col(2 * c2 + (row64 >> 5), r) * K)[f4];
    // ---- power-of-two operand scales from the factor maxima (k_absmax partials) and max|Y|; uniform (see k_grad_f16_v8) ----
    float scA, scS, scR, unP, unA, unS;
    {
        float* red = reinterpret_cast<float*>(smem);
        float m0 = 0.f, m1 = 0.f;
        for (int i = tid; i < V8_NPART; i += V5_THREADS) { m0 = fmaxf(m0, a.absmax[i]); m1 = fmaxf(m1, a.absmax[V8_NPART + i]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
        if (lane == 0) { red[w] = m0; red[8 + w] = m1; }
        __syncthreads();
        float mA = red[0], mS = red[8];
        for (int i = 1; i < 8; ++i) { mA = fmaxf(mA, red[i]); mS = fmaxf(mS, red[8 + i]); }
        __syncthreads();                     // red aliases Sl
        if (f16_range_fault((float)K * mA * mS, a.ymax, a.rangeRatio, a.doA, a.doS, a.wstatus, tid)) return;
        int qA = 0, qS = 0, qR = 0;
        (void)frexpf(mA, &qA);
        (void)frexpf(mS, &qS);
        (void)frexpf(a.ymax + (float)K * mA * mS, &qR);
        const int eA = mA > 0.f ? 14 - qA : 0, eS = mS > 0.f ? 14 - qS : 0, eR = 14 - qR;
        scA = ldexpf(1.f, eA); scS = ldexpf(1.f, eS); scR = ldexpf(1.f, eR);
        unP = ldexpf(1.f, -(eA + eS)); unA = ldexpf(1.f, -(eR + eS)); unS = ldexpf(1.f, -(eR + eA));
    }
    {   // ---- all S terms of the region, once: 512 threads = 64 image rows x 8 float4 = TWO blocks per pass (a row of S^T is 32 floats)
        const int st_off = r * ROWB + ((((f4 >> 1) ^ v3_swz(r)) & 7) << 4) + 8 * (f4 & 1);
#pragma unroll
        for (int c2 = 0; c2 < NCB / 2; ++c2) {
            unsigned char* d = smem + (2 * c2 + (row64 >> 5)) * SLB + st_off;
            if constexpr (R3) {
                const float x[4] = {sr[c2].x, sr[c2].y, sr[c2].z, sr[c2].w};
                f16x4 t0, t1, t2;
#pragma unroll
                for (int q = 0; q < 4; ++q) { _Float16 h_, l_, m_; v8_split3(x[q], scS, h_, l_, m_); t0[q] = h_; t1[q] = l_; t2[q] = m_; }
                *reinterpret_cast<f16x4*>(d) = t0;
                *reinterpret_cast<f16x4*>(d + V5_S_TERM) = t1;
                *reinterpret_cast<f16x4*>(d + 2 * V5_S_TERM) = t2;
            } else {
                f16x4 t0, t1;
                v8_split2(sr[c2], scS, t0, t1);
                *reinterpret_cast<f16x4*>(d) = t0;
                *reinterpret_cast<f16x4*>(d + V5_S_TERM) = t1;
            }
        }
    }

    if (producer) {
        k1_set_priority(-a.consPrio);      // (PMX_K1_PRIO < 0: the producers instead -- A/B only)
        // ================================ producers: P = A S and R ================================================
        f32x16 p0, p1;
        f32x16 q0, q1;                       // R3: the small products' accumulators
        float yv[2][2][16];                  // Y in flight: [pair set][block of the pair][row i of the tile] (accumulator layout)
        float4 areg[KS][2];
        f16x8 afr[KS][NT];
        const int jw = __builtin_amdgcn_readfirstlane(j);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        unsigned yoff[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) yoff[i] = ((unsigned)((i & 3) + 8 * (i >> 2) + 4 * hi) * (unsigned)a.ldY + 2u * (unsigned)l31) * 4u;
        const float* ybase0 = a.Y + (int64_t)(row0 + jw * 32) * a.ldY + col0;
        auto pair_base = [&](int q) {        // pair q = blocks 2 q, 2 q + 1 (clamped past the end of the region)
            int brp = q >> 2;
            if (brp >= nrp) brp = nrp - 1;
            return reinterpret_cast<const char*>(ybase0 + (int64_t)brp * V5_BM * a.ldY + (q & 3) * 64);
        };
        auto load_pair_rows = [&](const char* base, auto set_c, auto i0_c, auto n_c) {
            constexpr int set = decltype(set_c)::value, i0 = decltype(i0_c)::value, n = decltype(n_c)::value;
#pragma unroll
            for (int i = i0; i < i0 + n; ++i) {
                asm volatile("" : "+v"(yoff[i]));       // (saddr-form requests: scalar base + one 32-bit lane offset; see k_grad_f16_v8)
                const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(base + yoff[i]));
                yv[set][0][i] = v[0];
                yv[set][1][i] = v[1];
            }
        };
        auto load_A = [&](int prow) {
            const float4* src = reinterpret_cast<const float4*>(a.A + (int64_t)(prow + j * 32 + l31) * K + hi * 8);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                areg[ks][0] = src[ks * 4];
                areg[ks][1] = src[ks * 4 + 1];
            }
        };
        auto make_afr = [&]() {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float x[8] = {areg[ks][0].x, areg[ks][0].y, areg[ks][0].z, areg[ks][0].w,
                                    areg[ks][1].x, areg[ks][1].y, areg[ks][1].z, areg[ks][1].w};
                if constexpr (R3) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { _Float16 h_, l_, m_; v8_split3(x[q], scA, h_, l_, m_); afr[ks][0][q] = h_; afr[ks][1][q] = l_; afr[ks][2][q] = m_; }
                } else {
                unsigned hh[4], ll[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v8_split_pair(x[2 * q], x[2 * q + 1], scA, hh[q], ll[q]);
                afr[ks][0] = __builtin_bit_cast(f16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
                afr[ks][1] = __builtin_bit_cast(f16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
                }
            }
        };
        const int pa0 = (j * 32 + l31) * ROWB + ((hi ^ v3_swz(j * 32 + l31)) << 4);   // chunk 2 ks + hi: ^ (ks << 5)
        auto publish_A = [&]() {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                *reinterpret_cast<f16x8*>(smem + OFF_A + (pa0 ^ (ks << 5))) = afr[ks][0];
                *reinterpret_cast<f16x8*>(smem + OFF_A + V5_A_TERM + (pa0 ^ (ks << 5))) = afr[ks][1];
            }
        };
        const int s_g1 = l31 * ROWB + ((hi ^ v3_swz(l31)) << 4);                 // P's B operand: row l31, chunk 2 ks + hi: ^ (ks << 5)
        const int r_w = l31 * 256 + (((4 * j) ^ v4_swz(l31)) << 4) + 8 * hi;      // R producer, ^ (g << 4)
        using yes = std::integral_constant<bool, true>;
        using no = std::integral_constant<bool, false>;
        using set0 = std::integral_constant<int, 0>;
        load_A(row0);
        load_pair_rows(pair_base(0), set0{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
#pragma unroll
        for (int i = 0; i < 16; ++i) { p1[i] = 0.f; q0[i] = 0.f; q1[i] = 0.f; yv[1][1][i] = 0.f; }      // the zero "block -1" of slot 0 (see k_grad_f16_v8)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();        // Sl published

        auto slot = [&](int rp, auto cb_c, f32x16& pc, f32x16& pp, f32x16& lc, f32x16& lp, auto gemm_c, auto epi_c) {
            constexpr int cb = decltype(cb_c)::value;
            constexpr bool GEMM = decltype(gemm_c)::value, EPI = decltype(epi_c)::value;
            constexpr int pset = ((cb + 7) >> 1) & 1, ptile = (cb + 7) & 1;     // pair set and place in its pair of block s - 1
            if constexpr (cb == 2 && GEMM) {         // block s-2 opened this row panel: the consumers start on it in this slot
                publish_A();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (GEMM && cb == 0) {         // block s opens a row panel: its A terms, then the next panel's rows
                make_afr();
                load_A(row0 + (rp + 1 < nrp ? rp + 1 : nrp - 1) * V5_BM);
            }
            f16x8 sv[KS][NT];
            if constexpr (GEMM) {
                const unsigned char* Slb = smem + cb * SLB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int so = s_g1 ^ (ks << 5);
                    sv[ks][0] = *reinterpret_cast<const f16x8*>(Slb + so);
                    sv[ks][1] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                    if constexpr (R3) sv[ks][2] = *reinterpret_cast<const f16x8*>(Slb + so + 2 * V5_S_TERM);
                }
            }
            unsigned char* Rb = smem + OFF_R + ((cb + 1) & 1) * V5_R_BYTES;    // block s - 1 has the other parity
            unsigned h2[4][2], l2[4][2];
            const char* ybase_n = nullptr;
            if constexpr (EPI && (cb & 1) == 0) ybase_n = pair_base(rp * 4 + (cb >> 1) + 1);      // blocks s + 2, s + 3
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                if constexpr (GEMM && R3) {
                    if (t < 5 * KS) {        // per k step: al sh, ah sl, a3 sh, ah s3 into the small products' accumulator, ah sh into the other
                        const int ks = t / 5, wh = t % 5;
                        if (wh == 4) {
                            f32x16 cin = pc;
                            if (t == 4) {
#pragma unroll
                                for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                            }
                            pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][0], sv[ks][0], cin, 0, 0, 0);
                        } else {
                            f32x16 cin = lc;
                            if (t == 0) {
#pragma unroll
                                for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                            }
                            lc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][wh == 0 ? 1 : (wh == 2 ? 2 : 0)], sv[ks][wh == 1 ? 1 : (wh == 3 ? 2 : 0)], cin, 0, 0, 0);
                        }
                    }
                } else if constexpr (GEMM) {
                    if (t < 3 * KS) {
                        const int ks = t / 3, wh = t % 3;        // al sh, ah sl, ah sh
                        f32x16 cin = pc;
                        if (t == 0) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) cin[i] = 0.f;
                        }
                        pc = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks][wh == 0 ? 1 : 0], sv[ks][wh == 1 ? 1 : 0], cin, 0, 0, 0);
                    }
                }
                if constexpr (EPI) {
                    if (t < 8) {
                        const int g = t >> 1, hf = t & 1;
                        float r[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int e = 4 * g + 2 * hf + q;
                            r[q] = pp[e] * unP - yv[pset][ptile][e];
                            if constexpr (R3) r[q] = __builtin_fmaf(lp[e], unP, r[q]);     // (P_hh - Y) + P_lo
                            if constexpr (LOSS) lossAcc += r[q] * r[q];
                        }
                        v8_split_pair(r[0], r[1], scR, h2[g][hf], l2[g][hf]);
                        if (hf == 1) {
                            const int o = r_w ^ (g << 4);
                            *reinterpret_cast<uint2*>(Rb + o) = make_uint2(h2[g][0], h2[g][1]);
                            *reinterpret_cast<uint2*>(Rb + V5_R_TERM + o) = make_uint2(l2[g][0], l2[g][1]);
                        }
                    } else if constexpr ((cb & 1) == 0) {
                        if (t == 8) load_pair_rows(ybase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
                        if (t == 9) load_pair_rows(ybase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
                        if (t == 10) load_pair_rows(ybase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
                        if (t == 11) load_pair_rows(ybase_n, std::integral_constant<int, pset>{}, std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            __builtin_amdgcn_s_barrier();
        };
        using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
        using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
        using c4 = std::integral_constant<int, 4>; using c5 = std::integral_constant<int, 5>;
        using c6 = std::integral_constant<int, 6>; using c7 = std::integral_constant<int, 7>;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            slot(rp, c0{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c1{}, p1, p0, q1, q0, yes{}, yes{});
            slot(rp, c2{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c3{}, p1, p0, q1, q0, yes{}, yes{});
            slot(rp, c4{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c5{}, p1, p0, q1, q0, yes{}, yes{});
            slot(rp, c6{}, p0, p1, q0, q1, yes{}, yes{});
            slot(rp, c7{}, p1, p0, q1, q0, yes{}, yes{});
        }
        slot(nrp, c0{}, p0, p1, q0, q1, no{}, yes{});
        slot(nrp, c1{}, p1, p0, q1, q0, no{}, no{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ================================ consumers: gA and gSt of block s-2 ======================================
        k1_set_priority(a.consPrio);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();              // Sl published

        f32x16 accS[2];                      // gSt of THIS wave's two blocks of the region (2 j, 2 j + 1), all 128 rows of every panel
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) accS[c][i] = 0.f;
        f32x16 accA;                         // gA: rows 32 j .. of the panel, the one 32-wide k tile
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[i] = 0.f;
        int r_t0, r_t1;                      // gA's A operand (R, transposing read)
        {
            const int m = j * 32 + 16 * (lq & 1) + 4 * (li & 3);
            const int n0 = 8 * hi + (li >> 2), n1 = n0 + 4;
            r_t0 = n0 * 256 + ((((m >> 3) ^ v4_swz(n0)) & 15) << 4) + 8 * ((m >> 2) & 1);
            r_t1 = n1 * 256 + ((((m >> 3) ^ v4_swz(n1)) & 15) << 4) + 8 * ((m >> 2) & 1);
        }
        auto tr_src = [&](int row, int k0) {
            const int kk = k0 + 16 * (lq & 1) + 4 * (li & 3);
            return row * ROWB + ((((kk >> 3) ^ v3_swz(row)) & 7) << 4) + 8 * ((kk >> 2) & 1);
        };
        const int s_t0 = tr_src(8 * hi + (li >> 2), 0), s_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);   // gA's B operand (k tile 0)
        const int r_g3 = l31 * 256 + ((hi ^ v4_swz(l31)) << 4);                                // gSt's A operand: rows 16 ks ..: ^ (ks << 5)
        const int a_t0 = tr_src(8 * hi + (li >> 2), 0), a_t1 = tr_src(8 * hi + 4 + (li >> 2), 0);   // gSt's B operand, + ks * 16 * ROWB
        auto flush_gA = [&](int prow) {
            float* p0_ = a.slabA + (int64_t)colRegion * M * K + (int64_t)(prow + j * 32 + 4 * hi) * K + l31;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float* ph_ = p0_ + half * 16 * K;
                asm volatile("" : "+v"(ph_));          // keep it ONE pointer: the offsets below fold into the store's immediate
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = half * 8 + q;        // tile_row(i) = (i & 3) + 8 * (i >> 2) + 4 * hi
                    ph_[((q & 3) + 8 * (q >> 2)) * K] = accA[i] * unA;
                }
            }
        };
        auto sync = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        };
        auto consume = [&](int b, int prow, int cb, f32x16& accSc) {     // block b: column block cb of the panel at row prow
            const unsigned char* Rb = smem + OFF_R + (b & 1) * V5_R_BYTES;
            const unsigned char* Slb = smem + cb * SLB;
            const unsigned char* Ab = smem + OFF_A;
            if (a.doA & 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {           // the block's 32 columns: two steps of 16
                    const f16x8 r0 = v8_tr_pair(Rb, r_t0 + ks * 4096, r_t1 + ks * 4096);
                    const f16x8 r1 = v8_tr_pair(Rb + V5_R_TERM, r_t0 + ks * 4096, r_t1 + ks * 4096);
                    const int so0 = s_t0 + ks * 16 * ROWB, so1 = s_t1 + ks * 16 * ROWB;
                    const f16x8 s00 = v8_tr_pair(Slb, so0, so1);
                    const f16x8 s01 = v8_tr_pair(Slb + V5_S_TERM, so0, so1);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, s00, accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s01, accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, s00, accA, 0, 0, 0);
                }
            }
            if (a.doS && (cb >> 1) == j) {                 // (wave-uniform) this wave's block: all 128 rows of the panel, eight steps of 16
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
                    const f16x8 a0 = v8_tr_pair(Ab, ao0, ao1);
                    const f16x8 a1 = v8_tr_pair(Ab + V5_A_TERM, ao0, ao1);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r1, a0, accSc, 0, 0, 0);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a1, accSc, 0, 0, 0);
                    accSc = __builtin_amdgcn_mfma_f32_32x32x16_f16(r0, a0, accSc, 0, 0, 0);
                }
            }
            if ((a.doA & 1) && cb + 1 == NCB) {
                flush_gA(prow);
#pragma unroll
                for (int i = 0; i < 16; ++i) accA[i] = 0.f;
            }
        };
        sync();
        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int prow = row0 + rp * V5_BM;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (cb == 0) {               // block s-2 opens a row panel: the producers publish its A terms now
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                }
                consume(s - 2, prow, cb, accS[cb & 1]);
                sync();
                ++s;
            }
        }
        if (a.doS) {
            float* dst = a.slabS + (int64_t)rowRegion * N * K;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gn = block_col(2 * j + c, tile_row(i, lane));
                    dst[(int64_t)gn * K + l31] = accS[c][i] * unS;
                }
            }
        }
    }
    {
        float v = lossAcc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[w] = v;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < 4; ++i) s += (double)red[i];
            a.lossPart[blockIdx.x] = s;
        }
    }
}

// host side -----------------------------------------------------------------------------------------
// shapes the kernel takes (PMX_K1_K32=0: off -- the split-bf16 kernels of rounds 1-3 instead; tuning A/B)
bool grad_f16_k32_applies(int64_t M, int64_t N, int64_t K) {
    if (K != 32 || M % V5_BM != 0 || N % (V5_NB * V5_BN) != 0) return false;
    return !(getenv("PMX_K1_K32") && atoi(getenv("PMX_K1_K32")) == 0);
}
GradPlan grad_plan_f16_k32(int64_t M, int64_t N) {
    GradPlan p{};
    p.KP = 32;
    p.BN = V5_BN;
    const int64_t panels = M / V5_BM;
    p.gridY = (int)(N / (V5_NB * V5_BN));
    const int wantWG = getenv("PMX_K1_WGS") ? atoi(getenv("PMX_K1_WGS")) : 256;   // one resident workgroup per CU
    plan_row_regions(panels, p.gridY, wantWG, &p.RP, &p.gridX);
    p.nSlabA = p.gridY;
    p.nSlabS = p.gridX;
    p.ldsBytes = V8_LDS_BYTES;
    return p;
}
template <bool LOSS, bool R3 = false>
static hipError_t grad_launch_f16_k32_t(const GradV4Args& a, hipStream_t stream) {
    constexpr int lds = R3 ? V7_LDS_BYTES : V8_LDS_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)k_grad_f16_k32<LOSS, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_grad_f16_k32<LOSS, R3>), dim3(a.gridX * a.gridY), dim3(V5_THREADS), lds, stream, a);
    return hipGetLastError();
}
hipError_t grad_launch_f16_k32(const GradV4Args& a, hipStream_t stream) {
    if (a.r3) return (!(a.doA & 1) && !a.doS) ? grad_launch_f16_k32_t<true, true>(a, stream) : grad_launch_f16_k32_t<false, true>(a, stream);
    if (!(a.doA & 1) && !a.doS) return grad_launch_f16_k32_t<true>(a, stream);      // the loss-only pass (pmx_loglike, the line search)
    return grad_launch_f16_k32_t<false>(a, stream);
}
