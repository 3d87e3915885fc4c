"""GPU: the parity statement at the north star's own number.

north_star (BASELINE.json): "A, S matching the NumPy reference to rtol = 1e-4".  This file asserts exactly that --
|x - x_ref| <= 1e-5 + 1e-4 |x_ref| -- on EVERY entry of the factors for the back-ends whose arithmetic is smooth (pgm,
fista, adam / nadam, bsdmm) in the exact-fp32 mode, and states (and asserts) the fraction of entries that meet it for the
schemes whose Psi = sqrt(max(V, eps)) clamp (algorithms.py:181-183: amsgrad, padam, adamx) amplifies a summation-order
difference on a near-zero gradient entry by up to 1 / sqrt(eps) = 10^4.  It also runs the oracle at BASELINE's FULL sizes
(cfg3, cfg5; cfg4 on one GPU) for a few iterations -- the host needs ~1-4 s per fp64 iteration there -- compares RAdam's
early iterates (the fixture's later ones overflow in the reference itself), and the bSDMM Z / U buffers with the oracle's.

Measured fractions are written to gpurun_out/parity_fractions.json (copied to profiles/ when they are quoted)."""
import json
import os
from functools import partial

import numpy as np
import pytest

from conftest import as_spec, load_golden

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5                      # the north star's tolerance
SMOOTH = ("pgm", "fista", "adam", "nadam", "bsdmm")
REPORT = {}


def frac_within(actual, desired, rtol=RTOL, atol=ATOL):
    err = np.abs(np.asarray(actual, dtype=np.float64) - np.asarray(desired, dtype=np.float64))
    return float((err <= atol + rtol * np.abs(desired)).mean()), float((err / (atol + rtol * np.abs(desired))).max())


@pytest.fixture(scope="module")
def pm():
    import __graft_entry__ as g
    g.build()
    import proxmin_amd
    proxmin_amd.set_default_mode("f32")
    yield proxmin_amd
    proxmin_amd.set_default_mode(None)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_fractions.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


def _device_case(pm, c, Y, A0, S0, max_iter, e_rel):
    from test_gpu_nmf import run_device_case
    return run_device_case(pm, c, Y, A0, S0, max_iter, e_rel)


@pytest.mark.parametrize("fname", ["nmf_64x96_k8_f32.npz", "nmf_33x47_k3_f64.npz", "nmf_200x1000_k5_f64.npz"])
def test_fixtures_at_rtol_1e4_in_f32_mode(pm, orc, fname):
    """Reference-generated fixtures (25 iterations): smooth back-ends meet rtol 1e-4 on every entry; eps-clamp schemes on
    >= 99.5 % of them (the reference's own fp32 run against its fp64 run shows the same tail)."""
    z, meta = load_golden(fname)
    for name, c in meta["cases"].items():
        if name == "radam":
            continue                      # see test_radam_early_iterates
        tag = "unity" if c["unity_S"] else "plain"
        if "inputs_%s/Y" % tag in z.files:
            Y, A0, S0 = z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
        else:
            Y, A0, S0 = orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.dtype(meta["dtype"]).type, c["unity_S"], meta["seed"])
        A, S, _ = _device_case(pm, c, Y, A0, S0, meta["max_iter"], meta["e_rel"])
        fA, wA = frac_within(A, z[name + "/A"])
        fS, wS = frac_within(S, z[name + "/S"])
        REPORT["fixture %s %s" % (fname, name)] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS)}
        smooth = name.startswith(SMOOTH) and "unity" not in name    # prox_unity has no zero guard: a kink of its own
        if smooth:
            assert fA == 1.0 and fS == 1.0, "%s %s: %.5f / %.5f of the entries within rtol 1e-4 (worst %.1f x)" % (fname, name, fA, fS, max(wA, wS))
        else:
            assert fA >= 0.995 and fS >= 0.995, "%s %s: %.5f / %.5f" % (fname, name, fA, fS)


MEDIUM = [
    ("pgm", dict(), 1024, 1536, 32, False),
    ("fista", dict(accelerated=True), 1024, 1536, 32, False),
    ("adam", dict(scheme="adam"), 777, 1290, 64, False),
    ("bsdmm", dict(), 1024, 1024, 64, False),
    ("amsgrad_unity", dict(scheme="amsgrad"), 1536, 2048, 64, True),
]


@pytest.mark.parametrize("name,kw,M,N,K,unity", MEDIUM)
def test_medium_problems_at_rtol_1e4_in_f32_mode(pm, orc, name, kw, M, N, K, unity):
    """BASELINE-shaped problems at a size the fp64 oracle does in a second, 6 iterations from identical fp32 inputs."""
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=21)
    A, S = A0.copy(), S0.copy()
    Ao, So = A0.astype(np.float64), S0.astype(np.float64)
    Y64 = Y.astype(np.float64)
    ops = pm.operators
    if name in ("pgm", "fista"):
        step = pm.nmf.scaled_step_pgm(0.5) if name == "fista" else None
        ostep = (lambda a, s, it, g: tuple(0.5 * x for x in orc.lipschitz_steps(a, s))) if name == "fista" else None
        pm.nmf.nmf(Y, A, S, max_iter=6, e_rel=1e-9, step=step, **kw)
        orc.pgm_nmf(Y64, Ao, So, max_iter=6, e_rel=1e-9, step=ostep, **kw)
    elif name == "bsdmm":
        pgl = [[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2
        pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=6, e_rel=1e-9)
        orc.bsdmm_nmf(Y64, Ao, So, proxs_g=[[("plus",), ("soft", 1e-3, "relative")]] * 2, max_iter=6, e_rel=1e-9)
    else:
        pS = partial(ops.prox_unity_plus, axis=0) if unity else ops.prox_plus
        pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, prox_S=pS, max_iter=6, e_rel=1e-3, check_convergence=False, **kw)
        orc.adaprox_nmf(Y64, Ao, So, ("plus",), ("unity_plus", 0) if unity else ("plus",), max_iter=6, e_rel=1e-3,
                        check_convergence=False, **kw)
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    REPORT["medium %s %dx%dx%d" % (name, M, N, K)] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS)}
    if name.startswith(SMOOTH):
        assert fA == 1.0 and fS == 1.0, "%s: %.6f / %.6f within rtol 1e-4 (worst %.1f x)" % (name, fA, fS, max(wA, wS))
    else:
        assert fA >= 0.9999 and fS >= 0.999, (fA, fS)


SPLIT_MEDIUM = [
    # v8-shaped (K = 64, M % 128 = 0, N % 256 = 0) and k128-shaped (K = 128, M, N % 128 = 0) problems: the kernels the bench runs
    ("pgm", dict(), 1024, 1536, 64, False),
    ("fista", dict(accelerated=True), 1024, 1536, 64, False),
    ("adam", dict(scheme="adam"), 1024, 1536, 64, False),
    ("bsdmm", dict(), 1024, 1024, 64, False),
    ("pgm", dict(), 1024, 1024, 128, False),
    ("fista", dict(accelerated=True), 1024, 1024, 128, False),
    ("adam", dict(scheme="adam"), 1024, 1024, 128, False),
    ("bsdmm", dict(), 1024, 1024, 128, False),
    ("amsgrad_unity", dict(scheme="amsgrad"), 1536, 2048, 64, True),
]


def _solve_pair(pm, orc, name, kw, Y, A0, S0, unity, odtype):
    """the device run (current default mode) and the oracle run in `odtype` from identical fp32 inputs: 6 iterations"""
    A, S = A0.copy(), S0.copy()
    Ao, So, Yo = A0.astype(odtype), S0.astype(odtype), Y.astype(odtype)
    ops = pm.operators
    dev = pm is not None
    if name in ("pgm", "fista"):
        step = pm.nmf.scaled_step_pgm(0.5) if name == "fista" else None
        ostep = (lambda a, s, it, g: tuple(0.5 * x for x in orc.lipschitz_steps(a, s))) if name == "fista" else None
        pm.nmf.nmf(Y, A, S, max_iter=6, e_rel=1e-9, step=step, **kw)
        orc.pgm_nmf(Yo, Ao, So, max_iter=6, e_rel=1e-9, step=ostep, **kw)
    elif name == "bsdmm":
        pgl = [[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2
        pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=6, e_rel=1e-9)
        orc.bsdmm_nmf(Yo, Ao, So, proxs_g=[[("plus",), ("soft", 1e-3, "relative")]] * 2, max_iter=6, e_rel=1e-9)
    else:
        pS = partial(ops.prox_unity_plus, axis=0) if unity else ops.prox_plus
        pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, prox_S=pS, max_iter=6, e_rel=1e-3, check_convergence=False, **kw)
        orc.adaprox_nmf(Yo, Ao, So, ("plus",), ("unity_plus", 0) if unity else ("plus",), max_iter=6, e_rel=1e-3,
                        check_convergence=False, **kw)
    return A, S, Ao, So


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f16x2r"])
@pytest.mark.parametrize("name,kw,M,N,K,unity", SPLIT_MEDIUM)
def test_medium_problems_at_rtol_1e4_in_split_modes(pm, orc, name, kw, M, N, K, unity, mode):
    """[r4] The north-star number in the BENCH's own arithmetic: the split-precision modes on the shapes
    their tuned kernels take (k_grad_f16_v8 / k_grad_bf16_v7 at K = 64, k_grad_f16_k128 at K = 128), 6 iterations against the
    fp64 oracle from identical fp32 inputs.  Smooth back-ends (pgm, fista, adam, bsdmm): EVERY entry within
    |x - x_ref| <= 1e-5 + 1e-4 |x_ref|.  amsgrad + prox_unity_plus (cfg3's combination): the fraction within the bound is
    recorded next to the yardstick's -- the oracle itself in fp32 against the oracle in fp64 -- and held to a floor."""
    if mode == "bf16x3" and K == 128:
        pytest.skip("mode bf16x3 has no K = 128 kernel (the context runs the exact-fp32 kernel: covered by the f32 test)")
    from proxmin_amd.engine import DeviceNMF
    with DeviceNMF(M, N, K, mode=mode) as dev:
        kernel = dev.k1_info()["kernel"]
    want = ({"f16x2": "k_grad_f16_k128", "f16x2r": "k_grad_f16_k128_hh"}[mode] if K == 128 else
            {"f16x2": "k_grad_f16_v8", "bf16x3": "k_grad_bf16", "f16x2r": "k_grad_f16_v8_hh"}[mode])
    assert kernel == want, kernel
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=21)
    pm.set_default_mode(mode)
    try:
        A, S, Ao, So = _solve_pair(pm, orc, name, kw, Y, A0, S0, unity, np.float64)
    finally:
        pm.set_default_mode(None)
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    rec = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS), "kernel": kernel}
    if not name.startswith(SMOOTH):
        # yardstick: fp32 arithmetic in another summation order (the oracle in fp32) against the same fp64 run
        A32, S32 = A0.copy(), S0.copy()
        orc.adaprox_nmf(Y, A32, S32, ("plus",), ("unity_plus", 0) if unity else ("plus",), max_iter=6, e_rel=1e-3,
                        check_convergence=False, **kw)
        yA, ywA = frac_within(A32, Ao)
        yS, ywS = frac_within(S32, So)
        rec.update({"yardstick_frac_A": yA, "yardstick_frac_S": yS, "yardstick_worst_ratio": max(ywA, ywS),
                    "out_of_tolerance_vs_yardstick_A": (1.0 - fA) / max(1.0 - yA, 1.0 / A.size),
                    "out_of_tolerance_vs_yardstick_S": (1.0 - fS) / max(1.0 - yS, 1.0 / S.size)})
    REPORT["medium[%s] %s %dx%dx%d" % (mode, name, M, N, K)] = rec
    if name.startswith(SMOOTH):
        assert fA == 1.0 and fS == 1.0, "%s %s: %.6f / %.6f within rtol 1e-4 (worst %.1f x)" % (mode, name, fA, fS, max(wA, wS))
    else:
        if mode == "f16x2r":                 # the bench's arithmetic: tied to the yardstick (<= 2.5 x its out-of-tolerance count + 2 entries, <= 4 x its worst ratio)
            assert (1.0 - fA) <= 2.5 * (1.0 - yA) + 2.0 / A.size and (1.0 - fS) <= 2.5 * (1.0 - yS) + 2.0 / S.size, rec
            assert max(wA, wS) <= 4.0 * max(ywA, ywS, 1.0), rec
        else:
            assert fA >= 0.9995 and fS >= 0.998, rec


def test_radam_early_iterates(pm, orc):
    """RAdam (algorithms.py:224-245): rho <= 4 in the first iterations means Psi = 1 and a step of alpha * M / (1 - b1^t)
    -- with nmf()'s step sizes that overshoots, the iterates grow by orders of magnitude per iteration and overflow in the
    REFERENCE's own run.  While they are finite the device follows the reference's recorded iterates."""
    for fname in ("nmf_33x47_k3_f64.npz",):      # (the fp32 fixture has no radam case: NaN in the reference's own fp32 run)
        z, meta = load_golden(fname)
        c = meta["cases"]["radam"]
        tag = "unity" if c["unity_S"] else "plain"
        Y, A0, S0 = z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
        tb = pm.utils.Traceback()
        from test_gpu_nmf import run_device_case
        run_device_case(pm, c, Y, A0, S0, 6, meta["e_rel"], callback=tb)
        checked = 0
        for i in range(min(6, len(tb.trace))):
            if "radam/trace_A_%d" % i not in z.files:
                break
            rA, rS = z["radam/trace_A_%d" % i], z["radam/trace_S_%d" % i]
            if not (np.isfinite(rA).all() and np.isfinite(rS).all()) or max(np.abs(rA).max(), np.abs(rS).max()) > 1e30:
                break
            for got, want in ((tb.trace[i][0], rA), (tb.trace[i][1], rS)):
                scale = float(np.abs(want).max())
                np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * max(scale, 1.0), err_msg="%s radam iterate %d" % (fname, i))
            checked += 1
        assert checked >= 3, "only %d radam iterates could be compared" % checked


def test_bsdmm_Z_and_U_match_the_oracle(pm, orc):
    """utils.update_variables / do_the_mm (utils.py:295-346): the split variables Z_i and scaled duals U_i the device keeps
    per constraint, after 1 and after 4 iterations, against the oracle's (pinned by helpers.npz on the CPU side)."""
    from proxmin_amd import engine, operators as ops, _lib
    M, N, K = 300, 420, 12
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=3)
    specs = [("plus",), ("soft", 0.02, "relative")]
    for its in (1, 4):
        Ao, So = A0.astype(np.float64), S0.astype(np.float64)
        state = {}
        orc.bsdmm_nmf(Y.astype(np.float64), Ao, So, proxs_g=[specs, specs], max_iter=its, e_rel=1e-12, state=state)
        with engine.DeviceNMF(M, N, K, mode="f32") as dev:
            dev.set_Y(Y)
            dev.set_factors(A0, S0)
            pg = [ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_soft, thresh=0.02), 0)]
            dev.bsdmm_begin([ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(ops.prox_plus, 1)], [pg, pg], e_rel=(1e-12, 1e-12))
            assert dev.bsdmm_run(its).iterations == its
            for j, rows in ((0, M), (1, N)):
                for i in range(2):
                    Zd = dev._download(_lib.BUF_Z0 + j * _lib.MAX_G + i, rows)
                    Ud = dev._download(_lib.BUF_U0 + j * _lib.MAX_G + i, rows)
                    Zo, Uo = state["Z"][j][i], state["U"][j][i]
                    if j == 1:
                        Zd, Ud = Zd.T, Ud.T
                    np.testing.assert_allclose(Zd, Zo, rtol=1e-4, atol=1e-5, err_msg="Z block %d constraint %d after %d" % (j, i, its))
                    np.testing.assert_allclose(Ud, Uo, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(Uo).max())), err_msg="U block %d constraint %d after %d" % (j, i, its))


def _full_size(cfg, seed=4321):
    import torch
    import bench
    M, N, K, backend, unity, _ = bench.CONFIGS[cfg]
    Yd, A0, S0 = bench.make_problem_device(M, N, K, unity, seed, torch.device("cuda", 0))
    return M, N, K, backend, unity, Yd, A0, S0


@pytest.mark.parametrize("mode", ["f32", "f16x2", "f16x2r"])
def test_full_size_cfg3_against_the_oracle(orc, mode):
    """BASELINE cfg3 at its full size (16384 x 16384, K = 64, adaprox / AMSGrad, prox_plus + prox_unity_plus on the
    columns of S): 3 iterations against the fp64 oracle on the very same Y (copied back from the GPU), both the library's
    default arithmetic and the bench's.  Sub-iteration counts must agree exactly; factors per the trajectory policy."""
    import bench
    from proxmin_amd.engine import DeviceNMF
    M, N, K, backend, unity, Yd, A0, S0 = _full_size("cfg3")
    with DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Yd.data_ptr(), ld=N, copy=False, keepalive=Yd)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, backend, unity)
        r = run(3)
        A, S = dev.get_factors()
        sub = [int(r.sub_iterations[0]), int(r.sub_iterations[1])]
        if mode != "f32":
            assert dev.k1_info()["chain"] == 16 and dev.k1_info()["tail_fused"]
            assert dev.k1_info()["kernel"] == ("k_grad_f16_v8" if mode == "f16x2" else "k_grad_f16_v8_hh")
    Y64 = Yd.cpu().numpy().astype(np.float64)
    del Yd
    Ao, So = A0.astype(np.float64), S0.astype(np.float64)
    ret = orc.adaprox_nmf(Y64, Ao, So, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=3, e_rel=1e-3, check_convergence=False)
    assert sub == [int(ret[5][0]), int(ret[5][1])], (sub, ret[5])
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    REPORT["full cfg3 %s, 3 iterations vs fp64 oracle" % mode] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS), "sub_iterations": sub}
    # measured (profiles/r02_parity_fractions.json): f32 0.999999 / 0.99985, f16x2 0.99989 / 0.99946 -- AMSGrad's eps clamp.  The bench's mode
    # (f16x2r) is held to exact fp32's floor here and to the yardstick rule in tests/test_gpu_parity_long.py
    floor = (0.9999, 0.9995) if mode in ("f32", "f16x2r") else (0.9995, 0.999)
    assert fA >= floor[0] and fS >= floor[1], (fA, fS)
    np.testing.assert_allclose(S.sum(0), 1.0, rtol=1e-5)


@pytest.mark.parametrize("mode", ["f16x2", "f16x2r"])
def test_full_size_cfg5_against_the_oracle(orc, mode):
    """BASELINE cfg5 (16384 x 16384, K = 64, bSDMM, prox_plus + prox_soft per factor): 2 iterations against the fp64
    oracle; smooth arithmetic: rtol 1e-4 on every entry, in the bench's arithmetic mode (f16x2r) and in f16x2."""
    import bench
    from proxmin_amd.engine import DeviceNMF
    M, N, K, backend, unity, Yd, A0, S0 = _full_size("cfg5")
    with DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Yd.data_ptr(), ld=N, copy=False, keepalive=Yd)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, backend, unity)
        assert run(2).iterations == 2
        A, S = dev.get_factors()
    Y64 = Yd.cpu().numpy().astype(np.float64)
    del Yd
    Ao, So = A0.astype(np.float64), S0.astype(np.float64)
    orc.bsdmm_nmf(Y64, Ao, So, proxs_g=[[("plus",), ("soft", 1e-3, "relative")]] * 2, max_iter=2, e_rel=1e-12)
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    REPORT["full cfg5 %s, 2 iterations vs fp64 oracle" % mode] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS)}
    assert fA == 1.0 and fS == 1.0, (fA, fS, wA, wS)


_CFG4_ORACLE = {}


@pytest.mark.parametrize("mode", ["f32", "f16x2", "f16x2r"])
def test_full_size_cfg4_rows_of_one_rank_against_the_oracle(orc, mode):
    """BASELINE cfg4's per-GPU share (8192 of its 65536 rows x 16384, K = 128, adaprox / AMSGrad, prox_plus): 2 iterations
    on one GPU against the fp64 oracle, in the exact-fp32 mode and in the two-term fp16 mode (k_grad_f16_k128)."""
    import torch
    import bench
    from proxmin_amd.engine import DeviceNMF
    _, N, K, backend, unity, _ = bench.CONFIGS["cfg4"]
    M = 8192
    Yd, A0, S0 = bench.make_problem_device(M, N, K, unity, 77, torch.device("cuda", 0))
    with DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["kernel"] == {"f32": "k_grad_f32", "f16x2": "k_grad_f16_k128", "f16x2r": "k_grad_f16_k128_hh"}[mode]
        dev.set_Y_device(Yd.data_ptr(), ld=N, copy=False, keepalive=Yd)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, backend, unity)
        assert run(2).iterations == 2
        A, S = dev.get_factors()
    if not _CFG4_ORACLE:
        Y64 = Yd.cpu().numpy().astype(np.float64)
        Ao, So = A0.astype(np.float64), S0.astype(np.float64)
        orc.adaprox_nmf(Y64, Ao, So, ("plus",), ("plus",), scheme="amsgrad", max_iter=2, e_rel=1e-3, check_convergence=False)
        _CFG4_ORACLE["A"], _CFG4_ORACLE["S"] = Ao, So
    del Yd
    Ao, So = _CFG4_ORACLE["A"], _CFG4_ORACLE["S"]
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    REPORT["cfg4 share 8192x16384x128 %s, 2 iterations vs fp64 oracle" % mode] = {"frac_A": fA, "frac_S": fS, "worst_ratio": max(wA, wS)}
    assert fA >= 0.9999 and fS >= 0.9999, (fA, fS)
