"""Register-allocation guard for the hot kernels (CPU-side: hipcc cross-compiles gfx950 without a GPU).

A run-time `if (a.W != nullptr)` in the residual epilogue of k_grad_f32 once cost the UNWEIGHTED kernel ~100 spilled
VGPRs and 40 % of its speed without failing a single parity test.  This compiles the translation unit with
-Rpass-analysis=kernel-resource-usage and bounds the spill counts of the kernels the benchmark runs."""
import os
import re
import subprocess
import tempfile

import pytest

import __graft_entry__ as g

# mangled-name fragment -> max spilled VGPRs (current values in the comments)
LIMITS = {
    "13k_grad_f16_v8ILb0ELb0ELb1ELb0ELb0E": 8,    # 3   two-term fp16 K1 with the chained gA accumulation (bench default)
    "13k_grad_f16_v8ILb0ELb0ELb0ELb0ELb0E": 8,    # 0   the same with one gA slab per column region
    "13k_grad_f16_v8ILb0ELb0ELb0ELb1ELb0E": 8,    # 6   its loss-only instance
    # [r4] <.., R3> (mode f16x2r): a second accumulator and the third terms of A and S in the producers; one scratch access in the slot loop
    "13k_grad_f16_v8ILb0ELb0ELb1ELb0ELb1E": 12,   # 8   chained
    "13k_grad_f16_v8ILb0ELb0ELb0ELb0ELb1E": 12,   # 9   slabs
    "13k_grad_f16_v8ILb0ELb0ELb0ELb1ELb1E": 24,   # 18  loss-only
    # [r5] <.., HH> (mode f16x2r since round 5: the residual from the high x high product, the rest as a correction slab)
    "13k_grad_f16_v8ILb0ELb0ELb1ELb0ELb0ELb1E": 8,   # 4   chained (the bench's kernel)
    "13k_grad_f16_v8ILb0ELb0ELb0ELb0ELb0ELb1E": 8,   # 0   slabs
    # [r5] <.., HH, RS>: the consumers' roles split by contraction (the bench's kernel since then): gSt's waves hold the panel's A fragments in registers
    "13k_grad_f16_v8ILb0ELb0ELb1ELb0ELb0ELb1ELb1E": 4,   # 1   chained (256 VGPRs with the gSt waves' per-slot accumulator)
    "13k_grad_f16_v8ILb0ELb0ELb0ELb0ELb0ELb1ELb1E": 4,   # 2   slabs
    "13k_grad_f16_v8ILb0ELb0ELb0ELb0ELb0ELb1ELb0ELb1E": 0,   # 0   <ONLYS>: gSt alone, the wave's A fragments in registers (bsdmm's S step)
    "15k_grad_f16_k128ILb0ELb1ELb1ELb1E": 0,             # 0   K = 128, chained (the row split's chained instance spills 9)
    "15k_grad_f16_k128ILb0ELb0ELb1ELb1E": 0,             # 0   K = 128, slabs
    "11k_gfix_gram": 0,                   # 0   the correction's three kernels (k_gfix.hip)
    "12k_gfix_apply": 0,
    "13k_gfix_reduce": 0,
    "13k_grad_f16_v8ILb0ELb1ELb0E": 8,    # 0   weighted (5 in its loss-only instance)
    "13k_grad_f16_v8ILb0ELb1ELb1E": 8,    # 3   weighted, chained
    # two-term fp16 K1 at K = 128.  [r4] gSt re-split (one k tile per consumer wave, all 128 rows): 128 accumulator registers
    # in the consumers instead of 192 -- round 3's instances spilled 29 each, with reloads in the consumers' loop
    "15k_grad_f16_k128ILb0ELb0E": 4,      # 0   one gA slab per column region
    "15k_grad_f16_k128ILb1ELb0E": 8,      # 2   weighted
    "15k_grad_f16_k128ILb0ELb1E": 12,     # 10  chained gA (the previous sum arrives in two halves of 32 registers); every spill store
                                          #     sits in the prologue, every reload behind the loops (checked in the ISA: none inside)
    "15k_grad_f16_k128ILb1ELb1E": 16,     # 12  weighted, chained
    "14k_grad_f16_k32": 4,                # 0   [r4] two-term fp16 K1 at K = 32 (cfg2 in mode f16x2)
    "13k_grad_f32_pc": 4,                 # 0   exact-fp32 K1 with producer / consumer wavefronts (eight instances: 2 in the
                                          #     weighted, chained K = 64 one, 0 in the others)
    "10k_ada_tailILi2E": 0,               # 0   fused adaprox tail (K <= 64)
    "10k_ada_tailILi4E": 0,
    "14k_grad_bf16_v7ILb0ELb0E": 4,    # 0   split-bf16 K1 at K = 64
    "14k_grad_bf16_v7ILb0ELb1E": 20,   # 6   its weighted instance (17 when chained)
    "10k_grad_f32ILi64ELb0E": 32,      # 19  exact-fp32 K1, unweighted
    "10k_grad_f32ILi32ELb0E": 16,      # 3
    "10k_grad_f32ILi128ELb0E": 24,     # 11
}


def test_hot_kernels_do_not_spill():
    try:
        hipcc = g._hipcc()
    except RuntimeError:
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "--cuda-device-only", "-c",
               "-Rpass-analysis=kernel-resource-usage", os.path.join(g.CSRC, "pmx_api.hip"), "-o", os.path.join(tmp, "pmx.o")]
        r = subprocess.run(cmd, cwd=g.CSRC, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    spills, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and cur:
            spills[cur] = int(m.group(1))
    assert spills, "no resource-usage remarks in the compiler output"
    for frag, limit in LIMITS.items():
        hits = {k: v for k, v in spills.items() if frag in k}
        assert hits, "kernel %s not found in the build" % frag
        for k, v in hits.items():
            assert v <= limit, "%s spills %d VGPRs (limit %d)" % (k, v, limit)
