"""GPU: parity beyond a few iterations and at the full BASELINE sizes that the strict file leaves to bench lines
(task 4).

* the eps-clamp yardstick as a TEST: at full cfg3 the device's out-of-tolerance fraction against the fp64 oracle is bounded
  by what the reference's own arithmetic (the oracle run in fp32, NumPy / BLAS like the reference) shows against fp64;
* 20 iterations of cfg3: the bench's arithmetic (f16x2, chained K1, fused tail) against the library default (exact fp32),
  and the fp64 oracle on a 512-row x full-N problem over the same 20 iterations;
* the K = 128 kernel over 30 iterations against the exact-fp32 kernel (cfg4's 8192-row share);
* cfg2 end to end at its full 4096 x 4096 (PGM and damped FISTA, 10 iterations) against the fp64 oracle, every entry;
* cfg4 at its FULL size on one GPU (65536 x 16384, K = 128): one gradient pass against the row / column-subsampled fp64
  oracle.
Measured fractions go to gpurun_out/parity_long.json (copied to profiles/ when quoted)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5
REPORT = {}


def frac_within(actual, desired, rtol=RTOL, atol=ATOL):
    err = np.abs(np.asarray(actual, dtype=np.float64) - np.asarray(desired, dtype=np.float64))
    bound = atol + rtol * np.abs(desired)
    return float((err <= bound).mean()), float((err / bound).max())


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as g
    g.build()
    from proxmin_amd import engine
    yield engine
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_long.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


def _run_device(eng, mode, M, N, K, backend, unity, Yd, A0, S0, its, ld=None):
    import bench
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Yd.data_ptr(), ld=ld or N, copy=False, keepalive=Yd)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, backend, unity)
        r = run(its)
        assert r.iterations == its
        A, S = dev.get_factors()
        return A, S, [int(r.sub_iterations[0]), int(r.sub_iterations[1])], dev.k1_info()


def test_cfg3_full_size_device_error_against_the_references_own_fp32_error(eng, orc):
    """The AMSGrad eps-clamp story as a test.  Full cfg3 from a cold start, [r6] EIGHT iterations of both oracles with a snapshot
    after three: the fp64 oracle is the truth; the oracle run in fp32 (NumPy fp32 GEMMs: the reference's own arithmetic for fp32
    inputs, nmf.py:39-41) is the yardstick.
    After 3 iterations (the round-2..5 bars, unchanged except for the anchors):
    * mode f32 (exact fp32 MFMA): <= 2 x the yardstick's out-of-tolerance fraction (+ 2e-5) per block; worst entry <= 3 x the
      yardstick's worst OF THE SAME BLOCK with an absolute floor of 100 x the bound ([r6] ADVICE r5: the anchor used to be the
      larger block's worst, behind which a regression in A could hide; A's own yardstick worst is a single entry 2 % over the
      bound, hence the floor): measured A 82 / S 62.
    * [r5] the library's default and the bench's arithmetic (mode f16x2r: k_grad_f16_v8<HH> + k_gfix.hip): <= 2.5 x the yardstick's
      fraction (+ 2e-5), worst <= 4 x the block's yardstick worst (floor 200 x the bound); [r6] and in ENTRIES of A, where the
      yardstick has one: <= 3 x mode f32's count + 4.  The out-of-tolerance entries of A are LISTED (index, error over bound, the oracle's
      second moment V, first moment M and A there) for the yardstick and for modes f32 and f16x2r.  Measured (profiles/r06_*_parity_long.json):
      yardstick 1 entry, mode f32 5, mode f16x2r 13 of 1 048 576 -- every one of them between 1.0 and 4.4 x the bound, NOT at AMSGrad's
      eps clamp (V between 2e-4 and 1.7 there), clustered by ROW of A (rows 7048, 7830, 10511 carry 2-4 each; 4 of mode f32's 5 are also in
      mode f16x2r's list): rows whose gradient is a small difference of large terms three iterations after a cold start, where
      M / sqrt(V) ~ sign(g) turns a relative gradient error into a relative error of the step.  After EIGHT iterations no entry of A is out
      of tolerance in any arithmetic (worst 0.83 x the bound in mode f16x2r, 0.60 x in mode f32, 0.36 x for the yardstick).
    * modes f16x2 / bf16x3: absolute floors only -- they are NOT claimed to meet the yardstick.
    After 8 iterations: the same comparison recorded, modes f32 and f16x2r asserted at <= 3 x the yardstick's fraction (+ 5e-5)
    and the same pass counts as one of the two oracles."""
    import torch
    import bench
    M, N, K, backend, unity, _ = bench.CONFIGS["cfg3"]
    Yd, A0, S0 = bench.make_problem_device(M, N, K, unity, 4321, torch.device("cuda", 0))
    dev_out, dev_out8 = {}, {}
    for mode in ("f32", "f16x2", "bf16x3", "f16x2r"):     # (bf16x3: recorded beside the others -- three bf16 terms in A S, two in the gradient products)
        dev_out[mode] = _run_device(eng, mode, M, N, K, backend, unity, Yd, A0, S0, 3)
    for mode in ("f32", "f16x2r"):
        dev_out8[mode] = _run_device(eng, mode, M, N, K, backend, unity, Yd, A0, S0, 8)
    Y32 = Yd.cpu().numpy()
    del Yd

    def oracle_run(dtype):
        A, S = A0.astype(dtype), S0.astype(dtype)
        Mm, Vv = [np.zeros_like(A), np.zeros_like(S)], [np.zeros_like(A), np.zeros_like(S)]
        snap = {}

        def cb(A_, S_, it=None):
            if it == 3:
                snap.update(A=A_.copy(), S=S_.copy(), V_A=Vv[0].copy(), M_A=Mm[0].copy())

        ret = orc.adaprox_nmf(Y32.astype(dtype) if dtype != np.float32 else Y32, A, S, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=8,
                              e_rel=1e-3, check_convergence=False, M=Mm, V=Vv, callback=cb, sub_trace=True)
        per = ret[6]
        return snap, (A, S), [int(sum(p[j] for p in per[:3])) for j in range(2)], [int(sum(p[j] for p in per)) for j in range(2)]

    snap64, fin64, sub64_3, sub64_8 = oracle_run(np.float64)
    snap32, fin32, sub32_3, sub32_8 = oracle_run(np.float32)
    del Y32
    A64, S64 = snap64["A"], snap64["S"]
    yard = {"A": frac_within(snap32["A"], A64), "S": frac_within(snap32["S"], S64)}
    REPORT["cfg3 full, 3 its: oracle fp32 vs oracle fp64 (yardstick)"] = {"frac_A": yard["A"][0], "frac_S": yard["S"][0], "worst_ratio": max(yard["A"][1], yard["S"][1]),
                                                                        "worst_ratio_A": yard["A"][1], "worst_ratio_S": yard["S"][1],
                                                                        "sub_iterations_fp32": sub32_3, "sub_iterations_fp64": sub64_3}

    def bad_entries_of_A(A):
        err = np.abs(np.asarray(A, dtype=np.float64) - A64)
        bound = ATOL + RTOL * np.abs(A64)
        idx = np.argwhere(err > bound)
        return [[int(i), int(k), float(err[i, k] / bound[i, k]), float(snap64["V_A"][i, k]), float(snap64["M_A"][i, k]), float(A64[i, k])] for i, k in idx]

    listing = {"yardstick (oracle fp32)": bad_entries_of_A(snap32["A"]), "columns": ["row", "k", "error / bound", "V (fp64 oracle, after 3 its)", "M (same)", "A (same)"]}
    counts_A = {}
    for mode, (A, S, sub, info) in dev_out.items():
        assert sub == sub64_3, (mode, sub, sub64_3)
        got = {"A": frac_within(A, A64), "S": frac_within(S, S64)}
        rec = REPORT["cfg3 full, 3 its: device %s vs oracle fp64" % mode] = {"frac_A": got["A"][0], "frac_S": got["S"][0], "worst_ratio": max(got["A"][1], got["S"][1]),
                                                                           "worst_ratio_A": got["A"][1], "worst_ratio_S": got["S"][1]}
        if mode in ("f32", "f16x2r"):
            listing[mode] = bad_entries_of_A(A)
            counts_A[mode] = len(listing[mode])
        for b in ("A", "S"):
            out_dev, out_ref = 1.0 - got[b][0], 1.0 - yard[b][0]
            rec["out_of_tolerance_over_yardstick_%s" % b] = out_dev / max(out_ref, 1e-9)
            if mode == "f32":
                assert out_dev <= 2.0 * out_ref + 2e-5, (mode, b, out_dev, out_ref)
                assert got[b][1] <= 3.0 * max(yard[b][1], 100.0 / 3.0), (mode, b, got[b][1], yard[b][1])
            elif mode == "f16x2r":
                # the headline's arithmetic, in exact fp32's class: [r4, <R3>] measured A 8 entries of 1 M (yardstick 1, mode f32 5), S 1.8 x the
                # yardstick (mode f32 1.1 x, mode f16x2 4.2 x), worst entry 136 x the bound (82 x / 905 x); [r5, <HH>]: profiles/r05_*_parity_long.json
                assert info["kernel"] == "k_grad_f16_v8_hh", info
                assert out_dev <= 2.5 * out_ref + 2e-5, (mode, b, out_dev, out_ref)
                assert got[b][1] <= 4.0 * max(yard[b][1], 50.0), (mode, b, got[b][1], yard[b][1])
            else:
                assert out_dev <= (2.5e-4 if b == "A" else 1e-3), (mode, b, out_dev, out_ref)
                assert got[b][1] <= 2000.0, (mode, b, got[b][1])
    # [r6] A in ENTRIES (1 048 576 of them; the yardstick has ~1 out of tolerance): the bench's arithmetic tied to exact fp32's count
    n_yard = len(listing["yardstick (oracle fp32)"])
    assert counts_A["f16x2r"] <= max(3 * counts_A["f32"], 3 * n_yard) + 4, (counts_A, n_yard)
    eps_floor = 1e-8                          # AMSGrad's eps (algorithms.py:176-177: Psi = sqrt(max(V, eps)) in the reference's cold-start semantics)
    listing["shared f16x2r / f32"] = len({(e[0], e[1]) for e in listing["f16x2r"]} & {(e[0], e[1]) for e in listing["f32"]})
    listing["fraction of the listed entries with V below 100 eps"] = {m: (float(np.mean([e[3] < 100 * eps_floor for e in listing[m]])) if listing[m] else None) for m in ("f32", "f16x2r")}
    REPORT["cfg3 full, 3 its: out-of-tolerance entries of A"] = listing
    # ---- 8 iterations ---------------------------------------------------------------------------------------------------------------------------
    A64, S64 = fin64
    yard8 = {"A": frac_within(fin32[0], A64), "S": frac_within(fin32[1], S64)}
    REPORT["cfg3 full, 8 its: oracle fp32 vs oracle fp64 (yardstick)"] = {"frac_A": yard8["A"][0], "frac_S": yard8["S"][0], "worst_ratio_A": yard8["A"][1], "worst_ratio_S": yard8["S"][1],
                                                                        "sub_iterations_fp32": sub32_8, "sub_iterations_fp64": sub64_8}
    for mode, (A, S, sub, info) in dev_out8.items():
        got = {"A": frac_within(A, A64), "S": frac_within(S, S64)}
        rec = REPORT["cfg3 full, 8 its: device %s vs oracle fp64" % mode] = {"frac_A": got["A"][0], "frac_S": got["S"][0], "worst_ratio_A": got["A"][1], "worst_ratio_S": got["S"][1], "sub_iterations": sub}
        assert sub in (sub64_8, sub32_8), (mode, sub, sub64_8, sub32_8)
        for b in ("A", "S"):
            out_dev, out_ref = 1.0 - got[b][0], 1.0 - yard8[b][0]
            rec["out_of_tolerance_over_yardstick_%s" % b] = out_dev / max(out_ref, 1e-9)
            assert out_dev <= 3.0 * out_ref + 5e-5, (mode, b, out_dev, out_ref)


def test_cfg3_twenty_iterations_bench_mode_against_default_mode(eng):
    """20 iterations of full cfg3: f16x2 (chained k_grad_f16_v8 + fused tail) against exact fp32 (k_grad_f32_pc): the two
    K1s share no arithmetic.  Same proximal pass counts, S columns on the simplex, >= 99.99 % of A / 99.95 % of S within the
    north star's bound of each other (eps-clamp entries excepted); relative Frobenius distance < 2e-5 for A and < 2e-4 for S
    (measured 3.8e-6 / 8.4e-5: S carries the clamp entries, worst 94 x the bound)."""
    import torch
    import bench
    M, N, K, backend, unity, _ = bench.CONFIGS["cfg3"]
    Yd, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
    Af, Sf, subf, _ = _run_device(eng, "f32", M, N, K, backend, unity, Yd, A0, S0, 20)
    Ah, Sh, subh, info = _run_device(eng, "f16x2", M, N, K, backend, unity, Yd, A0, S0, 20)
    Ah2, Sh2, _, _ = _run_device(eng, "f16x2", M, N, K, backend, unity, Yd, A0, S0, 20)
    assert info["chain"] == 16 and info["tail_fused"] and info["chain_faults"] == 0
    assert np.array_equal(Ah, Ah2) and np.array_equal(Sh, Sh2), "f16x2 run is not repeatable bit for bit"
    assert subf == subh, (subf, subh)
    fA, wA = frac_within(Ah, Af)
    fS, wS = frac_within(Sh, Sf)
    relA = np.linalg.norm(Ah.astype(np.float64) - Af) / np.linalg.norm(Af)
    relS = np.linalg.norm(Sh.astype(np.float64) - Sf) / np.linalg.norm(Sf)
    REPORT["cfg3 full, 20 its: f16x2 vs f32 mode"] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS), "rel_frobenius": [float(relA), float(relS)], "sub_iterations": subh}
    assert fA >= 0.9999 and fS >= 0.9995, (fA, fS)
    assert relA < 2e-5 and relS < 2e-4, (relA, relS)
    np.testing.assert_allclose(Sh.sum(0), 1.0, rtol=1e-5)
    # mode f16x2r (the bench's: k_grad_f16_v8<HH> + the correction slab) over the same 20 iterations: bit-repeatable, same pass counts, and CLOSER to exact fp32 than f16x2 is
    Ar, Sr, subr, infor = _run_device(eng, "f16x2r", M, N, K, backend, unity, Yd, A0, S0, 20)
    Ar2, Sr2, _, _ = _run_device(eng, "f16x2r", M, N, K, backend, unity, Yd, A0, S0, 20)
    assert infor["kernel"] == "k_grad_f16_v8_hh" and infor["chain"] == 16 and infor["tail_fused"] and infor["chain_faults"] == 0
    assert np.array_equal(Ar, Ar2) and np.array_equal(Sr, Sr2), "f16x2r run is not repeatable bit for bit"
    assert subr == subf, (subr, subf)
    relAr = np.linalg.norm(Ar.astype(np.float64) - Af) / np.linalg.norm(Af)
    relSr = np.linalg.norm(Sr.astype(np.float64) - Sf) / np.linalg.norm(Sf)
    fAr, wAr = frac_within(Ar, Af)
    fSr, wSr = frac_within(Sr, Sf)
    REPORT["cfg3 full, 20 its: f16x2r vs f32 mode"] = {"frac_A": fAr, "frac_S": fSr, "worst_ratio": max(wAr, wSr), "rel_frobenius": [float(relAr), float(relSr)], "sub_iterations": subr}
    assert fAr >= 0.9999 and fSr >= 0.9995 and relAr < 2e-5 and relSr < 2e-4, (fAr, fSr, relAr, relSr)
    assert relSr <= relS and (1.0 - fSr) <= (1.0 - fS) + 1e-6, (relSr, relS, fSr, fS)


@pytest.mark.parametrize("mode", ["f32", "f16x2", "f16x2r"])
def test_cfg3_shape_rows_512_twenty_iterations_against_fp64_oracle(eng, orc, mode):
    """cfg3's solver on a 512-row x full-N problem (the fast K1 kernels take it: M % 128 = 0, N % 256 = 0), 20 iterations
    against the fp64 oracle: pass counts equal, >= 99.5 % of S and 99.9 % of A within rtol 1e-4."""
    import torch
    import bench
    M, N, K = 512, 16384, 64
    Yd, A0, S0 = bench.make_problem_device(M, N, K, True, 99, torch.device("cuda", 0))
    A, S, sub, info = _run_device(eng, mode, M, N, K, "adaprox", True, Yd, A0, S0, 20)
    assert info["kernel"] == {"f16x2": "k_grad_f16_v8", "f16x2r": "k_grad_f16_v8_hh", "f32": "k_grad_f32_pc"}[mode], info
    A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
    ret = orc.adaprox_nmf(Yd.cpu().numpy().astype(np.float64), A64, S64, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=20, e_rel=1e-3, check_convergence=False)
    assert sub == [int(ret[5][0]), int(ret[5][1])], (sub, ret[5])
    fA, wA = frac_within(A, A64)
    fS, wS = frac_within(S, S64)
    REPORT["512 x 16384 x 64 adaprox/amsgrad unity, 20 its: device %s vs oracle fp64" % mode] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS), "sub_iterations": sub}
    assert fA >= 0.999 and fS >= 0.995, (fA, fS)


@pytest.mark.parametrize("mode", ["f16x2", "f16x2r"])
@pytest.mark.parametrize("backend", ["pgm", "adaprox"])
def test_k128_kernel_thirty_iterations_against_exact_fp32(eng, backend, mode):
    """cfg4's 8192-row share, 30 iterations: k_grad_f16_k128 and [r5] k_grad_f16_k128<HH> + the correction slab (mode f16x2r: cfg4's
    arithmetic in bench.py), each twice (bit-identical), against k_grad_f32<128>."""
    import torch
    import bench
    M, N, K = 8192, 16384, 128
    Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    Af, Sf, subf, inf = _run_device(eng, "f32", M, N, K, backend, False, Yd, A0, S0, 30)
    Ah, Sh, subh, inh = _run_device(eng, mode, M, N, K, backend, False, Yd, A0, S0, 30)
    Ah2, Sh2, _, _ = _run_device(eng, mode, M, N, K, backend, False, Yd, A0, S0, 30)
    assert inh["kernel"] == ("k_grad_f16_k128" if mode == "f16x2" else "k_grad_f16_k128_hh") and inf["kernel"] == "k_grad_f32"
    assert np.array_equal(Ah, Ah2) and np.array_equal(Sh, Sh2)
    assert subf == subh
    fA, wA = frac_within(Ah, Af, rtol=2e-4, atol=2e-5)
    fS, wS = frac_within(Sh, Sf, rtol=2e-4, atol=2e-5)
    rel = max(np.linalg.norm(Ah.astype(np.float64) - Af) / np.linalg.norm(Af), np.linalg.norm(Sh.astype(np.float64) - Sf) / np.linalg.norm(Sf))
    REPORT["cfg4 share, 30 its %s: %s vs k_grad_f32<128>" % (backend, inh["kernel"])] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS), "rel_frobenius": float(rel)}
    assert fA == 1.0 and fS == 1.0, (fA, fS, wA, wS)
    assert rel < 5e-6, rel


@pytest.mark.parametrize("mode", ["f32", "f16x2", "f16x2r"])
@pytest.mark.parametrize("fista", [False, True])
def test_cfg2_end_to_end_at_full_size(eng, orc, fista, mode):
    """BASELINE cfg2 (4096 x 4096, K = 32, prox_plus, fp32): PGM and damped FISTA (step = 0.5 step_pgm),
    10 iterations through nmf() against the fp64 oracle: EVERY entry within rtol 1e-4 -- in the exact-fp32 mode BASELINE quotes
    the configuration in and [r4] in the two-term fp16 mode (k_grad_f16_k32)."""
    import torch
    import bench
    import proxmin_amd as pm
    M, N, K, _, _, _ = bench.CONFIGS["cfg2"]
    Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    Y = Yd.cpu().numpy()
    del Yd
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["kernel"] == {"f32": "k_grad_f32_pc", "f16x2": "k_grad_f16_k32", "f16x2r": "k_grad_f16_k32_r3"}[mode]
    pm.set_default_mode(mode)          # (f16x2r: the LIBRARY'S default since round 6 and the arithmetic bench.py quotes cfg2_f16x2r in)
    A, S = A0.copy(), S0.copy()
    kw = dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)) if fista else {}
    try:
        pm.nmf.nmf(Y, A, S, max_iter=10, e_rel=1e-12, **kw)
    finally:
        pm.set_default_mode(None)
    A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
    step = (lambda A_, S_, it, grads: tuple(0.5 * s_ for s_ in orc.lipschitz_steps(A_, S_))) if fista else None
    orc.pgm_nmf(Y.astype(np.float64), A64, S64, max_iter=10, e_rel=1e-12, accelerated=fista, step=step)
    fA, wA = frac_within(A, A64)
    fS, wS = frac_within(S, S64)
    REPORT["cfg2 full %s, 10 its vs fp64 oracle (%s mode)" % ("fista/2" if fista else "pgm", mode)] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS)}
    assert fA == 1.0 and fS == 1.0, (fA, fS, wA, wS)


@pytest.mark.parametrize("mode", ["f16x2", "f16x2r"])
def test_cfg4_full_size_gradient_against_subsampled_oracle(eng, mode):
    """BASELINE cfg4 at its FULL size on one GPU (65536 x 16384, K = 128; Y = 4 GiB): one pass of k_grad_f16_k128 (mode f16x2r: <HH> + the
    correction slab) over all of it, the gradients of 256 random rows of A and 256 random columns of S against the fp64 oracle."""
    import torch
    import bench
    from test_gpu_kernels import _subsampled_oracle_gradients
    M, N, K = 65536, 16384, 128
    Yd, A, S = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(M, 256, replace=False))
    cols = np.sort(rng.choice(N, 256, replace=False))
    rA, rS = _subsampled_oracle_gradients(A, S, Yd, rows, cols)
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["kernel"] == ("k_grad_f16_k128" if mode == "f16x2" else "k_grad_f16_k128_hh")
        dev.set_Y_device(Yd.data_ptr(), ld=N, copy=False, keepalive=Yd)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
    np.testing.assert_allclose(gA[rows], rA, rtol=2e-5, atol=2e-5 * np.abs(rA).max())
    np.testing.assert_allclose(gS[:, cols], rS, rtol=2e-5, atol=2e-5 * np.abs(rS).max())
    REPORT["cfg4 full 65536 x 16384 x 128 gradient vs subsampled fp64 oracle"] = {
        "max_err_gA_over_max": float(np.abs(gA[rows] - rA).max() / np.abs(rA).max()), "max_err_gS_over_max": float(np.abs(gS[:, cols] - rS).max() / np.abs(rS).max())}
