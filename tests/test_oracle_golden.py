"""CPU: pin the oracle (oracle/nmf_oracle.py) to the reference's outputs stored in tests/golden/."""
import numpy as np
import pytest

from conftest import as_spec, load_golden
from oracle import nmf_oracle as orc

TOL = {"float64": dict(rtol=1e-10, atol=1e-12), "float32": dict(rtol=2e-4, atol=2e-5)}


def run_oracle_case(c, Y, A0, S0, max_iter, e_rel, trace=None):
    A, S = A0.copy(), S0.copy()
    pA, pS = as_spec(c["prox_A"]), as_spec(c["prox_S"])
    if c["alg"] == "pgm":
        step = None
        if c["half_step"]:
            step = lambda a, s, it, g: tuple(0.5 * x for x in orc.lipschitz_steps(a, s))  # noqa: E731
        ret = orc.pgm_nmf(Y, A, S, pA, pS, step=step, max_iter=max_iter, e_rel=e_rel, trace=trace, **c["kw"])
    elif c["alg"] == "adaprox":
        ret = orc.adaprox_nmf(Y, A, S, pA, pS, max_iter=max_iter, e_rel=e_rel, trace=trace, **c["kw"])
    else:
        pg = c["proxs_g"]
        if pg is not None:
            pg = [None if g is None else [as_spec(s) for s in g] for g in pg]
        ret = orc.bsdmm_nmf(Y, A, S, pA, pS, proxs_g=pg, max_iter=max_iter, e_rel=e_rel, trace=trace)
    return A, S, ret


def _inputs(z, meta, unity):
    tag = "unity" if unity else "plain"
    key = "inputs_%s/Y" % tag
    if key in z.files:
        return z[key], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
    Y, A0, S0 = orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.dtype(meta["dtype"]).type, unity, meta["seed"])
    chk = z["inputs_%s/checksum" % tag]
    got = np.array([Y.sum(dtype=np.float64), A0.sum(dtype=np.float64), S0.sum(dtype=np.float64), float(Y[0, 0]), float(Y[-1, -1])])
    np.testing.assert_allclose(got, chk, rtol=1e-13)
    return Y, A0, S0


@pytest.mark.parametrize("fname", ["nmf_200x1000_k5_f64.npz", "nmf_33x47_k3_f64.npz", "nmf_64x96_k8_f32.npz"])
def test_nmf_cases_match_reference(fname):
    z, meta = load_golden(fname)
    tol = TOL[meta["dtype"]]
    for name, c in meta["cases"].items():
        Y, A0, S0 = _inputs(z, meta, c["unity_S"])
        trace = []
        A, S, ret = run_oracle_case(c, Y, A0, S0, meta["max_iter"], meta["e_rel"], trace)
        np.testing.assert_allclose(A, z[name + "/A"], err_msg=name, **tol)
        np.testing.assert_allclose(S, z[name + "/S"], err_msg=name, **tol)
        assert len(trace) == int(z[name + "/n_callbacks"]), name
        i = 0
        while "%s/trace_A_%d" % (name, i) in z.files:
            np.testing.assert_allclose(trace[i][0], z["%s/trace_A_%d" % (name, i)], err_msg=name, **tol)
            np.testing.assert_allclose(trace[i][1], z["%s/trace_S_%d" % (name, i)], err_msg=name, **tol)
            i += 1
        np.testing.assert_allclose(orc.half_sq_residual(A, S, Y), float(z[name + "/loss"]), rtol=max(tol["rtol"], 1e-9))
        if c["alg"] == "pgm":
            conv, G, steps, _ = ret
            np.testing.assert_allclose(G[0], z[name + "/G_A"], rtol=tol["rtol"] * 50, atol=tol["atol"] * 1e3)
            np.testing.assert_allclose(G[1], z[name + "/G_S"], rtol=tol["rtol"] * 50, atol=tol["atol"] * 1e3)
            np.testing.assert_allclose(np.array(steps, dtype=np.float64), z[name + "/steps"], rtol=max(tol["rtol"], 1e-9))
            assert list(conv) == list(z[name + "/conv"])
        elif c["alg"] == "adaprox":
            conv, Mm, Vv, Vh, _, _ = ret
            np.testing.assert_allclose(Mm[0], z[name + "/M_A"], rtol=tol["rtol"] * 10, atol=tol["atol"] * 1e2)
            np.testing.assert_allclose(Vv[1], z[name + "/V_S"], rtol=tol["rtol"] * 10, atol=tol["atol"] * 1e2)
            assert [v is None for v in Vh] == list(z[name + "/vhat_none"])
            assert [bool(x) for x in conv] == list(z[name + "/conv"])
        else:
            assert list(ret[0]) == list(z[name + "/conv"])


def test_survey_known_answer_table():
    """SURVEY.md section 4 / BASELINE.md section 4 numbers, 200x1000 K=5 fp64, 25 iterations."""
    z, meta = load_golden("nmf_200x1000_k5_f64.npz")
    table = {"pgm": (6504.642228, 468.7944695, 2268.884499), "fista_half": (2040.265577, 507.3059664, 2456.199794),
             "adam": (3516.863701, 514.1751181, 2422.761991), "amsgrad": (2699.256928, 546.2252238, 2280.047093),
             "amsgrad_unityS": (605.774982, 495.3925036, 1000.0), "bsdmm_none": (2691.561203, 467.8474546, 2663.011374),
             "bsdmm_plus_soft": (2691.7553, 468.532618, 2658.900448)}
    for name, (loss, sA, sS) in table.items():
        c = meta["cases"][name]
        Y, A0, S0 = _inputs(z, meta, c["unity_S"])
        A, S, _ = run_oracle_case(c, Y, A0, S0, 25, 1e-6)
        assert orc.half_sq_residual(A, S, Y) == pytest.approx(loss, rel=2e-9)
        assert A.sum() == pytest.approx(sA, rel=2e-9)
        assert S.sum() == pytest.approx(sS, rel=2e-9)


def test_operators_match_reference():
    z, meta = load_golden("operators.npz")
    for e in meta["entries"]:
        k = e["key"]
        X, step = z[k + "/X"], z[k + "/step"]
        step = float(step) if step.ndim == 0 else step
        out = orc.apply_prox(X.copy(), step, tuple(e["spec"]))
        assert out.dtype == X.dtype or e["spec"][0] in ("min", "max", "hard", "hard_plus", "soft", "soft_plus")
        tol = dict(rtol=1e-6, atol=1e-7) if X.dtype == np.float32 else dict(rtol=1e-13, atol=0)
        np.testing.assert_allclose(out, z[k + "/out"], err_msg=str(e), **tol)
    ap = meta["ap"]
    out = orc.apply_prox_sequence(z["ap/X"], ap["step"], [tuple(s) for s in ap["specs"]], ap["repeat"])
    np.testing.assert_allclose(out, z["ap/out"], rtol=1e-13)


def test_helpers_match_reference():
    z, meta = load_golden("helpers.npz")
    np.testing.assert_allclose(orc.nesterov_omegas(40), z["nesterov/omega"], rtol=1e-15)
    b1 = z["moments/b1"]
    for e in meta["moments"]:
        k = e["key"]
        G, M, V = z[k + "/G"], z[k + "/M0"].copy(), z[k + "/V0"].copy()
        Vh = z[k + "/Vh0"].copy() if e["vhat"] else None
        Phi, Psi = orc.moment_update(e["scheme"], e["it"], G, M, V, Vh, b1, 0.999, 1e-8, 0.25)
        tol = dict(rtol=2e-6, atol=1e-9) if G.dtype == np.float32 else dict(rtol=1e-13, atol=0)
        for got, want in ((M, "M1"), (V, "V1"), (Phi, "Phi"), (Psi, "Psi")):
            np.testing.assert_allclose(got, z["%s/%s" % (k, want)], err_msg=str(e), **tol)
        if e["vhat"]:
            np.testing.assert_allclose(Vh, z[k + "/Vh1"], **tol)
    A, S = z["steps/A"], z["steps/S"]
    np.testing.assert_allclose(np.array(orc.lipschitz_steps(A, S)), z["steps/pgm"], rtol=1e-12)
    aA, aS = orc.adaprox_steps(A, S)
    assert aA.shape == z["steps/ada_A"].shape and aS.shape == z["steps/ada_S"].shape
    np.testing.assert_allclose(aA, z["steps/ada_A"], rtol=1e-14)
    np.testing.assert_allclose(aS, z["steps/ada_S"], rtol=1e-14)
    for typ in (1, 2):
        bb = orc.BBStepper(kind=typ, init_r=0.1)
        for it in range(6):
            X = (z["bb%d/X_A_%d" % (typ, it)], z["bb%d/X_S_%d" % (typ, it)])
            Gs = (z["bb%d/G_A_%d" % (typ, it)], z["bb%d/G_S_%d" % (typ, it)])
            np.testing.assert_allclose(np.array(bb.step(X, it, Gs)), z["bb%d/steps" % typ][it], rtol=1e-12)


def test_update_variables_matches_reference():
    """utils.update_variables/do_the_mm/check_constraint_convergence via one bsdmm-style block
    update re-derived from the oracle's formulas."""
    z, meta = load_golden("helpers.npz")
    uv = meta["uv"]
    X, G = z["uv/X0"].copy(), z["uv/G"]
    Z = [z["uv/Z0_0"].copy(), z["uv/Z0_1"].copy()]
    U = [z["uv/U0_0"].copy(), z["uv/U0_1"].copy()]
    sf, sg = uv["step_f"], uv["step_g"]
    dX = np.sum([sf / sg[i] * (X - Z[i] + U[i]) for i in range(2)], axis=0)
    X[:] = orc.apply_prox((X - dX) - sf * G, sf, tuple(uv["prox_f"]))
    np.testing.assert_allclose(X, z["uv/X1"], rtol=1e-13)
    for i in range(2):
        Zn = orc.apply_prox(X + U[i], sg[i], tuple(uv["proxs_g"][i]))
        R = X - Zn
        Sd = -1 / sg[i] * (Zn - Z[i])
        U[i] += R
        np.testing.assert_allclose(Zn, z["uv/Z1_%d" % i], rtol=1e-13)
        np.testing.assert_allclose(U[i], z["uv/U1_%d" % i], rtol=1e-13)
        np.testing.assert_allclose(R, z["uv/R_%d" % i], rtol=1e-13, atol=1e-16)
        np.testing.assert_allclose(Sd, z["uv/Sd_%d" % i], rtol=1e-13, atol=1e-16)


def test_unmixing_known_answers():
    """examples/unmixing.py: pgm with backtracking + adaprox with constant steps, to convergence."""
    z, meta = load_golden("unmixing.npz")
    Y, A0, S0 = z["Y"], z["A0"], z["S0"]
    for r in meta["runs"]:
        A, S = A0.copy(), S0.copy()
        pA, pS = tuple(r["prox_A"]), tuple(r["prox_S"])
        if r["cfg"] is None:
            _, _, _, n = orc.pgm_nmf(Y, A, S, pA, pS, backtracking=True, e_rel=1e-4, max_iter=1000)
        else:
            sch, a = r["cfg"]
            _, _, _, _, n, _ = orc.adaprox_nmf(Y, A, S, pA, pS, step=lambda *x, a=a: (a, a), scheme=sch, e_rel=1e-4, max_iter=1000)
        assert n == r["iters"], r["key"]
        assert orc.half_sq_residual(A, S, Y) == pytest.approx(r["loss"], rel=1e-8), r["key"]
        np.testing.assert_allclose(A, z[r["key"] + "/A"], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_weighted_likelihood_matches_reference(tag):
    """W as an M x N array (nmf.py:13-41): loss, gradients, adaprox (whose step rule ignores W) and pgm with a user
    step reproduce the reference; pgm's default step rule raises like the reference does (`W == 1` on an array)."""
    z, meta = load_golden("weighted.npz")
    meta = meta["cases"][tag]
    Y, A0, S0, W = z[tag + "/Y"], z[tag + "/A0"], z[tag + "/S0"], z[tag + "/W"]
    tol = TOL[str(Y.dtype)]
    assert orc.half_sq_residual(A0, S0, Y, W) == pytest.approx(float(z[tag + "/loss0"]), rel=tol["rtol"])
    gA, gS = orc.residual_gradients(A0, S0, Y, W)
    np.testing.assert_allclose(gA, z[tag + "/gA0"], **tol)
    np.testing.assert_allclose(gS, z[tag + "/gS0"], **tol)
    sc = meta["s_const"]
    runs = {
        "amsgrad": lambda A, S: orc.adaprox_nmf(Y, A, S, scheme="amsgrad", check_convergence=False, max_iter=10, e_rel=1e-6, W=W),
        "adam_unityS": lambda A, S: orc.adaprox_nmf(Y, A, S, ("plus",), ("unity_plus", 0), scheme="adam", check_convergence=False,
                                                    max_iter=10, e_rel=1e-6, W=W),
        "pgm_const_step": lambda A, S: orc.pgm_nmf(Y, A, S, step=lambda a, s, it, g: (sc, sc), max_iter=10, e_rel=1e-6, W=W),
    }
    loose = dict(rtol=5e-3, atol=5e-4) if tag == "f32" else tol     # fp32 trajectories: see DESIGN.md section 2
    for name, run in runs.items():
        A, S = A0.copy(), S0.copy()
        run(A, S)
        key = "%s/%s" % (tag, name)
        np.testing.assert_allclose(A, z[key + "/A"], err_msg=key, **loose)
        np.testing.assert_allclose(S, z[key + "/S"], err_msg=key, **loose)
        assert orc.half_sq_residual(A, S, Y, W) == pytest.approx(float(z[key + "/loss"]), rel=1e-3 if tag == "f32" else 1e-9)
    assert meta["default_step_error"] == "ValueError"
    with pytest.raises(ValueError):
        orc.pgm_nmf(Y, A0.copy(), S0.copy(), max_iter=2, W=W)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_array_valued_steps_match_reference(tag):
    """algorithms.pgm with a user `step` that returns arrays (they broadcast against the blocks, algorithms.py:106-108;
    prox_soft_plus scales its threshold with them): fixture array_steps.npz, generated from the reference."""
    z, meta = load_golden("array_steps.npz")
    Y, A0, S0 = z[tag + "/Y"], z[tag + "/A0"], z[tag + "/S0"]
    vecA, vecS, fullA, rowS = (z["%s/%s" % (tag, k)] for k in ("vecA", "vecS", "fullA", "rowS"))
    soft = ("soft_plus", 0.05, "relative")
    runs = {
        "vectors_plus": ((vecA, vecS), ("plus",), ("plus",), False),
        "vectors_plus_fista": ((vecA, vecS), ("plus",), ("plus",), True),
        "full_row_soft": ((fullA, rowS), soft, soft, False),
        "scalar_and_vector": ((float(vecA[0]), vecS), ("plus",), soft, True),
    }
    assert sorted(runs) == meta["cases"][tag]["runs"]
    tol = TOL[str(Y.dtype)] if tag == "f64" else dict(rtol=5e-3, atol=5e-4)       # fp32 trajectories: see DESIGN.md section 2
    for name, (st, pA, pS, accel) in runs.items():
        A, S = A0.copy(), S0.copy()
        ret = orc.pgm_nmf(Y, A, S, pA, pS, step=lambda a, s, it, g, st=st: st, accelerated=accel, max_iter=10, e_rel=1e-6)
        key = "%s/%s" % (tag, name)
        np.testing.assert_allclose(A, z[key + "/A"], err_msg=key, **tol)
        np.testing.assert_allclose(S, z[key + "/S"], err_msg=key, **tol)
        np.testing.assert_allclose(ret[1][0], z[key + "/gA"], err_msg=key, rtol=max(tol["rtol"], 1e-8), atol=1e-3 if tag == "f32" else 1e-9)
        assert ret[3] == int(z[key + "/n_callbacks"])


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_backtracking_with_a_user_step_matches_reference(tag):
    """algorithms.pgm(backtracking=True) with a user `step` (multiples of the Lipschitz steps: the line search has to halve T):
    fixture bt_user_step.npz, generated from the reference."""
    z, meta = load_golden("bt_user_step.npz")
    Y, A0, S0 = z[tag + "/Y"], z[tag + "/A0"], z[tag + "/S0"]
    runs = {"x1.5": (1.5, False), "x2_with_grads": (2.0, False), "x1_fista": (1.0, True)}
    assert sorted(runs) == meta["cases"][tag]["runs"]
    tol = TOL[str(Y.dtype)] if tag == "f64" else dict(rtol=5e-3, atol=5e-4)
    for name, (fac, accel) in runs.items():
        A, S = A0.copy(), S0.copy()
        ret = orc.pgm_nmf(Y, A, S, step=lambda a, s, it, g, fac=fac: tuple(fac * v for v in orc.lipschitz_steps(a, s)), accelerated=accel,
                          backtracking=True, max_iter=15, e_rel=1e-6)
        key = "%s/%s" % (tag, name)
        np.testing.assert_allclose(A, z[key + "/A"], err_msg=key, **tol)
        np.testing.assert_allclose(S, z[key + "/S"], err_msg=key, **tol)
        assert ret[3] == int(z[key + "/n_callbacks"])
