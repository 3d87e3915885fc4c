// chain_link.h -- the hand-off protocol of the chained in-place accumulation of gA, ONE copy for the four K1s that carry it
// (k_grad_bf16_v7, k_grad_f16_v8, k_grad_f32_pc, k_grad_f16_k128; rounds 2-3 pasted it into each of them).
//
// The workgroups that own the same rows and consecutive column regions form a chain: chainL workgroups, multiples of 8 apart in
// dispatch order, i.e. on ONE XCD where workgroup b runs on XCD b % 8.  They add their contributions to a panel of gA IN PLACE
// in one slab, through that XCD's L2, one after the other in a fixed order (deterministic).  Member c visits its row panels
// rotated by `stride * c`, so at any time the members of a chain work on different panels and member c reaches a panel `stride`
// panel-times after member c - 1 did.  Per consumer wave (its tile of the panel):
//     open(panel)    which visitor of this panel am I (k), is there a previous sum to add (k > 0)
//     look()         request the arrival word           -- in front of a block's MFMAs
//     wait()         ... and look at it behind them: normally the predecessor is long done (no spin); a predecessor on ANOTHER XCD
//                    (the word carries its writer's XCC_ID), one that does not show up within 20 ms (workgroups not co-resident), or a
//                    chain somebody else has given up on -> fault()
//     [the kernel fetches the previous sum with sc1 loads -- served by L2, never by the CU's L1 -- and adds it to its accumulators]
//     flushed()      after the wave's plain stores of the summed tile: the arrival k + 1 is pending ...
//     publish()      ... and goes out at the top of the next panel behind s_waitcnt vmcnt(0) (the stores are in L2): relaxed agent-scope
//                    store.  No release fence, no write-back: the lines stay dirty in the L2 the members share.
// fault() reports through DevStatus (k1_fault, halt): the chain of kernels stops before anything is updated and the host repeats
// the iteration with one slab per column region for the rest of the context's life (pmx_api.hip: chain_fault_fallback).
// HIP promises neither the placement nor the co-residency this relies on for SPEED; correctness never assumes them.
#pragma once

// workgroup `lin` of a chained launch -> (chain, place in the chain, row region, column region); gx = row regions
__device__ __forceinline__ void chain_region_map(int lin, int L, int gx, int& chainId, int& chainPos, int& rowRegion, int& colRegion) {
    const int xcd = lin & 7, idx = lin >> 3;
    chainPos = idx % L;
    chainId = (idx / L) * 8 + xcd;
    rowRegion = chainId % gx;
    colRegion = (chainId / gx) * L + chainPos;
}

struct ChainLink {
    unsigned* cflags = nullptr;      // arrival words of this chain, this wave: [panel][4]
    unsigned* curFlag = nullptr;
    unsigned* pendFlag = nullptr;    // arrival to publish once this wave's stores of the panel have landed
    unsigned pendVal = 0, cwant = 0, cseen = 0, myxcc = 0;
    bool cadd = false;               // this panel has a previous sum to add (not the first visitor of the panel)
    bool cdead = false;              // a fault was seen: no more waiting, the launch's gA is discarded anyway
    const DevStatus* status = nullptr;
    DevStatus* wstatus = nullptr;
    int lane = 0;

    __device__ __forceinline__ void init(unsigned* chainFlags, int chainId, int nrp, int wave, const DevStatus* st, DevStatus* wst, int lane_) {
        cflags = chainFlags + (size_t)chainId * nrp * 4 + wave;
        myxcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // hwreg(HW_REG_XCC_ID, 0, 4)
        status = st;
        wstatus = wst;
        lane = lane_;
    }
    __device__ __forceinline__ void fault(int code) {
        if (lane == 0 && code > 0) {
            wstatus->k1_fault = code;
            wstatus->reason = HALT_ERROR;
            __threadfence();
            wstatus->halt = 1;
        }
        cadd = false;
        cdead = true;
    }
    __device__ __forceinline__ void publish() {
        if (pendFlag != nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(pendFlag, pendVal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::: "memory");   // nothing of the next panel moves in front of the arrival store
            pendFlag = nullptr;
        }
    }
    // this workgroup's place among the visitors of panel pnl, in time: members that reach it after wrapping around
    // (pnl + stride c >= nrp) come first
    __device__ __forceinline__ void open(int pnl, int chainPos, int L, int stride, int nrp, unsigned base, bool doA) {
        publish();                           // the previous panel's arrival (its stores have had a slot to land)
        const int c0 = (nrp - pnl + stride - 1) / stride;             // first member that reaches pnl after wrapping around
        const int nw = L - c0 > 0 ? L - c0 : 0;
        const int k = pnl + stride * chainPos >= nrp ? chainPos - c0 : chainPos + nw;
        cadd = doA && k > 0 && !cdead;
        cwant = base + (unsigned)k;
        curFlag = cflags + pnl * 4;
    }
    __device__ __forceinline__ void look() {
        if (cadd) cseen = __hip_atomic_load(curFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void wait() {
        if (!cadd) return;
        unsigned v = __builtin_amdgcn_readfirstlane(cseen);
        if ((v >> 4) != cwant) {
            const long long t0 = wall_clock64();          // 100 MHz
            for (int spins = 1; (v >> 4) != cwant; ++spins) {
                if ((spins & 63) == 0) {
                    if (chain_halted(status)) { fault(0); break; }              // somebody else gave up
                    if (wall_clock64() - t0 > 2000000) { fault(1); break; }     // 20 ms
                }
                __builtin_amdgcn_s_sleep(8);
                v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(curFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
        if (cadd && (v & 15u) != myxcc) fault(2);
        asm volatile("" ::: "memory");   // the sc1 loads of the previous sum stay behind the arrival check
    }
    __device__ __forceinline__ void flushed() {
        pendFlag = curFlag;
        pendVal = ((cwant + 1u) << 4) | myxcc;
    }
};
