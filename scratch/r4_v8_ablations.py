#!/usr/bin/env python3
"""Round 4, VERDICT r3 item 4: UPPER-BOUND ablations of the three levers DESIGN section 8.1 names for k_grad_f16_v8.

Each lever, fully built, would remove some work from a block; each ablation here removes AT LEAST that work by simply not
doing it (the results are WRONG -- these libraries are for timing only, never shipped): if even that buys < 2 %, the real
restructuring (which keeps the arithmetic and pays for it in registers / LDS) cannot buy more.

  a   consumer operand reuse: a 64 x 32 register-blocked gSt tile would read each R fragment once for two MFMAs.
      Ablation: the gSt contraction reads its R fragments (two b128 reads) only on every second 16-row step.
  b   one barrier per two blocks (four-deep R ring): ablation: both roles skip the barrier after every even block.
  c   producers computing two 32-row blocks per S-fragment read: ablation: the producers read the S fragments of steps 0, 1
      and use them for steps 2, 3 as well (half the reads per block).
  abc all three.

Builds scratch/libpmx_abl_<name>.so from patched copies of proxmin_amd/csrc under /tmp (the product sources are untouched)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "proxmin_amd", "csrc")

EDITS = {
    "a": [("""                    const int ro = r_g3 ^ (ks << 5);
                    const f16x8 r0 = *reinterpret_cast<const f16x8*>(Rb + ro);
                    const f16x8 r1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro);
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
                    const f16x8 a0 = v8_tr_pair(Ab, ao0, ao1);""",
           """                    const int ro = r_g3 ^ (ks << 5);
                    if ((ks & 1) == 0) { rr0 = *reinterpret_cast<const f16x8*>(Rb + ro); rr1 = *reinterpret_cast<const f16x8*>(Rb + V5_R_TERM + ro); }
                    const f16x8 r0 = rr0, r1 = rr1;
                    const int ao0 = a_t0 + ks * 16 * ROWB, ao1 = a_t1 + ks * 16 * ROWB;
                    const f16x8 a0 = v8_tr_pair(Ab, ao0, ao1);"""),
          ("""            if (a.doS) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);""",
           """            if (a.doS) {
                f16x8 rr0, rr1;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ro = r_g3 ^ (ks << 5);""")],
    "b": [("""            PH(3)
            PH(4)
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            __builtin_amdgcn_s_barrier();
            PH(0)""",
           """            PH(3)
            PH(4)
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes retired before the barrier
            if constexpr ((cb & 1) == 1) __builtin_amdgcn_s_barrier();
            PH(0)"""),
          ("""                PH(7)
                sync();
                ++s;""",
           """                PH(7)
                if (cb & 1) sync();
                ++s;"""),
          ("""        sync();
        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int pnl = panel_at(rp);""",
           """        sync();
        int s = 2;
#pragma nounroll
        for (int rp = 0; rp < nrp; ++rp) {
            const int pnl = panel_at(rp);""")],
    "c": [("""                for (int ks = 0; ks < 4; ++ks) {
                    const int so = s_g1 ^ (ks << 5);
                    sv[ks][0] = *reinterpret_cast<const f16x8*>(Slb + so);
                    sv[ks][1] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                }""",
           """                for (int ks = 0; ks < 2; ++ks) {
                    const int so = s_g1 ^ (ks << 5);
                    sv[ks][0] = *reinterpret_cast<const f16x8*>(Slb + so);
                    sv[ks][1] = *reinterpret_cast<const f16x8*>(Slb + so + V5_S_TERM);
                    sv[ks + 2][0] = sv[ks][0];
                    sv[ks + 2][1] = sv[ks][1];
                }""")],
}
EDITS["abc"] = EDITS["a"] + EDITS["b"] + EDITS["c"]


def build(name):
    d = "/tmp/abl_%s" % name
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d + "/proxmin_amd")
    shutil.copytree(SRC, d + "/proxmin_amd/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), d + "/include")
    f = d + "/proxmin_amd/csrc/k_grad_f16_v8.hip"
    s = open(f).read()
    for old, new in EDITS[name]:
        assert s.count(old) == 1, (name, old[:60], s.count(old))
        s = s.replace(old, new)
    open(f, "w").write(s)
    out = os.path.join(ROOT, "scratch", "libpmx_abl_%s.so" % name)
    return subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "pmx_api.hip", "-o", out, "-ldl"],
                            cwd=d + "/proxmin_amd/csrc", stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    procs = [(n, build(n)) for n in (sys.argv[1:] or ["a", "b", "c", "abc"])]
    for n, p in procs:
        print(n, "rc", p.wait())
