#!/usr/bin/env python3
"""Round 4: where does the chained K = 128 K1 pay its 8 % (0.372 against 0.343 ms with slabs)?  Ablation library: the SLAB kernel with
the chain's panel ROTATION (member c visits its panels rotated by c) but none of the protocol -- if the rotation alone costs the
difference, it is the locality of the A-term fragment loads (every workgroup of an XCD on another panel), not the hand-off."""
import os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/abl_rot"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d + "/proxmin_amd")
shutil.copytree(os.path.join(ROOT, "proxmin_amd", "csrc"), d + "/proxmin_amd/csrc")
shutil.copytree(os.path.join(ROOT, "include"), d + "/include")
f = d + "/proxmin_amd/csrc/k_grad_k128.hip"
s = open(f).read()
old = """    const int rot = CHAIN ? a.chainStride * chainPos : 0;
    auto panel_at = [&](int t) {
        if constexpr (CHAIN) { const int p = t - rot; return p < 0 ? p + nrp : p; }
        else return t;
    };"""
new = """    const int rot = CHAIN ? a.chainStride * chainPos : (colRegion & 15);
    auto panel_at = [&](int t) {
        const int p = t - rot; return p < 0 ? p + nrp : p;
    };"""
assert s.count(old) == 1
s = s.replace(old, new)
# the consumers of the slab instance must flush to the rotated panel as well: they already use panel_at(rp) for prow
open(f, "w").write(s)
out = os.path.join(ROOT, "scratch", "libpmx_abl_rot.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "pmx_api.hip", "-o", out, "-ldl"],
                      cwd=d + "/proxmin_amd/csrc", stderr=subprocess.DEVNULL)
print("built", out)
