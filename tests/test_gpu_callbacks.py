"""GPU: the reference's callback surface with callables that are NOT objects of this library:
`step(*X, it=None[, grads=None])` (algorithms.py:73-77, :370) and `prox(X, step) -> X'` (:37-39) written by the user run
through a host round trip, one iteration per call -- the reference's own idioms must work unchanged:
`step=lambda *X, it=None: tuple(0.5 * s for s in step_pgm(*X))` and `step=lambda *X, it: (alpha, alpha)`
(examples/unmixing.py:139-143)."""
import logging
from functools import partial

import numpy as np
import pytest

from conftest import as_spec, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["library-default", "f32"])
def pm(request):
    """[r6] the callback / user-callable surface in the LIBRARY'S default arithmetic (f16x2r: what a user who swaps the import gets)
    and in exact fp32"""
    import __graft_entry__ as g
    g.build()
    import proxmin_amd
    proxmin_amd.set_default_mode(None if request.param == "library-default" else request.param)
    if request.param == "library-default":
        assert proxmin_amd.get_default_mode() == proxmin_amd.LIBRARY_DEFAULT_MODE == "f16x2r"
    yield proxmin_amd
    proxmin_amd.set_default_mode(None)


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


def test_lambda_step_reproduces_the_fista_half_fixture(pm, caplog):
    """SURVEY.md section 4's `fista_half` row was generated from the reference with exactly this lambda."""
    from test_gpu_nmf import assert_factors_close
    for fname in ("nmf_200x1000_k5_f64.npz", "nmf_33x47_k3_f64.npz"):
        z, meta = load_golden(fname)
        c = meta["cases"]["fista_half"]
        from oracle import nmf_oracle as orc
        tag = "unity" if c["unity_S"] else "plain"
        if "inputs_%s/Y" % tag in z.files:
            Y, A0, S0 = z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
        else:
            Y, A0, S0 = orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.dtype(meta["dtype"]).type, c["unity_S"], meta["seed"])
        A, S = A0.copy(), S0.copy()
        with caplog.at_level(logging.WARNING, logger="proxmin"):
            ret = pm.nmf.nmf(Y, A, S, accelerated=True, step=lambda *X, it=None: tuple(0.5 * s for s in pm.nmf.step_pgm(*X)),
                             max_iter=meta["max_iter"], e_rel=meta["e_rel"])
        assert_factors_close(A, z["fista_half/A"], meta["dtype"], fname + " A")
        assert_factors_close(S, z["fista_half/S"], meta["dtype"], fname + " S")
        np.testing.assert_allclose(np.array(ret[2], dtype=np.float64), z["fista_half/steps"], rtol=1e-4)
        # and it equals the fused form of the same rule
        A2, S2 = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A2, S2, accelerated=True, step=pm.nmf.scaled_step_pgm(0.5), max_iter=meta["max_iter"], e_rel=meta["e_rel"])
        np.testing.assert_allclose(A, A2, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(S, S2, rtol=1e-5, atol=1e-6)


def test_lambda_learning_rate_for_adaprox_equals_constant_step(pm, orc):
    """examples/unmixing.py:139-143: `step=lambda *X, it: (alpha, alpha)`; the same numbers as the fused constant_step."""
    Y, A0, S0 = orc.synthetic_problem(300, 500, 12, np.float32, seed=9)
    out = []
    for step in (lambda *X, it: (0.01, 0.02), pm.nmf.constant_step(0.01, 0.02)):
        A, S = A0.copy(), S0.copy()
        ret = pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", step=step, max_iter=15, e_rel=1e-4,
                         prox_S=partial(pm.operators.prox_soft_plus, thresh=1e-3))
        out.append((A, S, ret))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert out[0][2][0] == out[1][2][0]
    # per-component arrays, the shapes nmf.step_adaprox itself returns
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", step=lambda *X, it=None: pm.nmf.step_adaprox(*X), max_iter=6, e_rel=1e-4)
    A2, S2 = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A2, S2, algorithm=pm.adaprox, scheme="adam", max_iter=6, e_rel=1e-4)
    np.testing.assert_array_equal(A, A2)
    np.testing.assert_array_equal(S, S2)


def test_unmixing_example_with_its_own_lambda(pm):
    """examples/unmixing.py:126-161 as written there: the solver called DIRECTLY with the example's own step lambda
    (`lambda *X, it: (alpha, alpha)`), e_rel 1e-4, prox_max_iter 100; the reference's final losses and iteration counts are
    in tests/golden/unmixing.npz (generated from the reference)."""
    z, meta = load_golden("unmixing.npz")
    Y, A0, S0 = z["Y"], z["A0"], z["S0"]
    rows = [r for r in meta["runs"] if r["mode"] == "nmf" and r["cfg"] is not None and r["cfg"][0] == "adam"]
    assert rows
    for r in rows:
        alpha = r["cfg"][1]
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        grad = partial(pm.nmf.grad_likelihood, Y=Y)
        pm.adaprox((A, S), grad, lambda *X, it: (alpha, alpha), prox=[pm.operators.prox_plus, pm.operators.prox_plus], max_iter=1000,
                   callback=tb, e_rel=1e-4, b1=0.9, b2=0.999, prox_max_iter=100, scheme="adam")
        loss = pm.nmf.log_likelihood(A, S, Y=Y)
        assert abs(loss / r["loss"] - 1) < 5e-3, (alpha, loss, r["loss"])
        # a converged non-convex run, fp32 against the reference's fp64: same basin and loss; the iteration at which the
        # 1e-4 stopping test fires moves with the rounding of the gradient sums (677 in the reference, 640-790 here depending
        # on which K1 runs)
        assert abs(len(tb.trace) - r["iters"]) <= 0.25 * r["iters"], (alpha, len(tb.trace), r["iters"])


def my_plus(X, step):
    """a user-written projection onto the non-negative numbers"""
    X[X < 0] = 0
    return X


def my_soft(X, step, thresh=1e-3):
    return np.sign(X) * np.maximum(np.abs(X) - thresh * step, 0)


@pytest.mark.parametrize("kw", [dict(), dict(accelerated=True, half=True)])
def test_user_written_prox_in_pgm(pm, orc, kw):
    Y, A0, S0 = orc.synthetic_problem(400, 640, 24, np.float32, seed=2)
    kw = dict(kw)
    step = pm.nmf.scaled_step_pgm(0.5) if kw.pop("half", False) else None
    runs = {}
    for name, pA, pS in (("lib", pm.operators.prox_plus, pm.operators.prox_plus), ("user", my_plus, my_plus), ("mixed", my_plus, pm.operators.prox_plus)):
        A, S = A0.copy(), S0.copy()
        ret = pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, step=step, max_iter=10, e_rel=1e-9, **kw)
        runs[name] = (A, S, ret)
    for name in ("user", "mixed"):
        np.testing.assert_array_equal(runs[name][0], runs["lib"][0])      # same arithmetic: v = Xe - s G, then the projection
        np.testing.assert_array_equal(runs[name][1], runs["lib"][1])
        np.testing.assert_array_equal(runs[name][2][1][0], runs["lib"][2][1][0])


def test_user_written_prox_in_adaprox(pm, orc):
    """the proximal loop of the block with a user-defined prox runs around the callable on the host: same iterates as the
    library operator up to fp32 rounding (numpy rounds the loop's expression twice where the device fuses a multiply-add),
    same sub-iteration counts; the callback sees every iterate; StopIteration ends the run."""
    Y, A0, S0 = orc.synthetic_problem(300, 420, 8, np.float32, seed=7)
    lib = partial(pm.operators.prox_soft, thresh=1e-3)
    res = {}
    for name, pS in (("lib", lib), ("user", my_soft)):
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        ret = pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="amsgrad", prox_A=pm.operators.prox_plus if name == "lib" else my_plus,
                         prox_S=pS, max_iter=8, e_rel=1e-3, callback=tb)
        res[name] = (A, S, ret, len(tb.trace))
    np.testing.assert_allclose(res["user"][0], res["lib"][0], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(res["user"][1], res["lib"][1], rtol=2e-4, atol=2e-5)
    assert res["user"][3] == res["lib"][3]
    np.testing.assert_allclose(res["user"][2][1][0], res["lib"][2][1][0], rtol=1e-3, atol=1e-6)     # M of block A

    def stopper(*X, it=None):
        if it == 3:
            raise StopIteration

    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, prox_S=my_soft, max_iter=8, e_rel=1e-3, callback=stopper)
    A3, S3 = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A3, S3, algorithm=pm.adaprox, prox_S=my_soft, max_iter=3, e_rel=1e-3)
    np.testing.assert_array_equal(A, A3)
    np.testing.assert_array_equal(S, S3)


def test_slow_path_warns_once(pm, orc, caplog):
    from proxmin_amd import algorithms
    algorithms._warned.clear()
    Y, A0, S0 = orc.synthetic_problem(64, 96, 4, np.float32, seed=1)
    with caplog.at_level(logging.WARNING, logger="proxmin"):
        for _ in range(2):
            A, S = A0.copy(), S0.copy()
            pm.nmf.nmf(Y, A, S, prox_A=my_plus, max_iter=2, e_rel=1e-9)
    msgs = [r.getMessage() for r in caplog.records if "user-defined Python callable" in r.getMessage()]
    assert len(msgs) == 1, msgs


def test_prox_unity_along_the_long_axis(pm, orc):
    """operators.prox_unity(axis=0) -- the reference's default axis (operators.py:41-52) -- normalises the COLUMNS of its
    argument: a sum over the long dimension of a factor.  Stand-alone it runs the column-sum + scale kernels; inside the
    solvers it is applied between kernel launches (one iteration per call) and must reproduce the oracle."""
    rng = np.random.default_rng(3)
    for shape, axis in (((700, 5), 0), ((5, 900), 1), ((300, 260), 0), ((4, 40000), 1)):
        X = (rng.random(shape) - 0.2).astype(np.float32)
        for fn, ref in ((pm.operators.prox_unity, lambda V: V / V.sum(axis=axis, keepdims=True)),
                        (pm.operators.prox_unity_plus, lambda V: np.maximum(V, 0) / np.maximum(V, 0).sum(axis=axis, keepdims=True))):
            got = fn(X.copy(), 1.0, axis=axis)
            np.testing.assert_allclose(got, ref(X.astype(np.float64)), rtol=2e-6, atol=1e-9)
    # inside nmf(): A's columns normalised (axis=0 on A), pgm and adaprox, against the oracle
    Y, A0, S0 = orc.synthetic_problem(260, 380, 6, np.float32, seed=5)
    A0 /= A0.sum(0, keepdims=True)
    for alg, kw, okw in ((pm.pgm, dict(), dict()), (pm.adaprox, dict(scheme="adam"), dict(scheme="adam"))):
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, algorithm=alg, prox_A=partial(pm.operators.prox_unity_plus, axis=0), max_iter=6, e_rel=1e-4, **kw)
        Ao, So = A0.astype(np.float64), S0.astype(np.float64)
        if alg is pm.pgm:
            orc.pgm_nmf(Y.astype(np.float64), Ao, So, prox_A=("unity_plus", 0), max_iter=6, e_rel=1e-4)
        else:
            orc.adaprox_nmf(Y.astype(np.float64), Ao, So, ("unity_plus", 0), ("plus",), max_iter=6, e_rel=1e-4, **okw)
        np.testing.assert_allclose(A, Ao, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(S, So, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(A.sum(0), 1.0, rtol=1e-5)
    # [r4] bsdmm with the LIBRARY's long-axis operator, as prox_f of A and as a constraint of S (rows of S: axis=1): the
    # operator runs as the stand-alone device kernel between the stages of pmx_bsdmm_split (round 3: NotImplementedError)
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, prox_A=partial(pm.operators.prox_unity_plus, axis=0),
               proxs_g=[[pm.operators.prox_plus], [partial(pm.operators.prox_unity_plus, axis=1)]], max_iter=5, e_rel=1e-9)
    Ao, So = A0.astype(np.float64), S0.astype(np.float64)
    orc.bsdmm_nmf(Y.astype(np.float64), Ao, So, prox_A=("unity_plus", 0), proxs_g=[[("plus",)], [("unity_plus", 1)]], max_iter=5, e_rel=1e-9)
    np.testing.assert_allclose(A, Ao, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(S, So, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(A.sum(0), 1.0, rtol=1e-5)


def test_bsdmm_update_order_and_direct_entry(pm, orc):
    """bsdmm's update_order (algorithms.py:730-736, :805) -- S before A, and a block left out -- and the direct entry
    algorithms.bsdmm(X, proxs_f, steps_f_cb, ...) with the closures nmf() itself builds."""
    Y, A0, S0 = orc.synthetic_problem(200, 300, 6, np.float32, seed=12)
    pgl = [[pm.operators.prox_plus, partial(pm.operators.prox_soft, thresh=0.01)]] * 2
    A, S = A0.copy(), S0.copy()
    conv = pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=5, e_rel=1e-9)
    A2, S2 = A0.copy(), S0.copy()
    pf, sf = pm.nmf.bsdmm_closures(Y, [pm.operators.prox_plus, pm.operators.prox_plus])
    conv2 = pm.bsdmm([A2, S2], pf, sf, proxs_g=pgl, max_iter=5, e_rel=1e-9)
    np.testing.assert_array_equal(A, A2)
    np.testing.assert_array_equal(S, S2)
    assert conv == conv2
    # generic closures run through the host path now (test_bsdmm_with_generic_closures); an identity prox_f leaves X alone
    A4, S4 = A0.copy(), S0.copy()
    pm.bsdmm([A4, S4], lambda X, step, Xs=None, j=None: X, lambda Xs, j=None: 1.0, max_iter=2)
    np.testing.assert_array_equal(A4, A0)
    # S first: equals the default order on the transposed problem (Y^T = S^T A^T)
    A3, S3 = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A3, S3, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=5, e_rel=1e-9, update_order=[1, 0])
    At, St = np.ascontiguousarray(S0.T), np.ascontiguousarray(A0.T)
    pm.nmf.nmf(np.ascontiguousarray(Y.T), At, St, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=5, e_rel=1e-9)
    np.testing.assert_allclose(S3, At.T, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(A3, St.T, rtol=1e-4, atol=1e-5)
    # only A is updated: S stays, its convergence flag stays None
    A4, S4 = A0.copy(), S0.copy()
    conv4 = pm.nmf.nmf(Y, A4, S4, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=3, e_rel=1e-9, update_order=[0])
    np.testing.assert_array_equal(S4, S0)
    assert conv4[1] is None and not np.array_equal(A4, A0)


def test_constant_step_with_user_prox_in_pgm(pm, orc):
    """pgm(step=nmf.constant_step(a, b), prox=<user callable>) ran the split phases with
    DevStatus::step == 0 (only pmx_pgm_run uploaded the constants): X did not move and the run "converged" at once.
    A user-written projection must reproduce the fused prox_plus run with the same constants."""
    Y, A0, S0 = orc.synthetic_problem(400, 640, 24, np.float32, seed=2)
    sA, sS = pm.nmf.step_pgm(A0, S0)
    step = pm.nmf.constant_step(0.5 * float(sA), 0.5 * float(sS))
    runs = {}
    for name, pA, pS in (("lib", pm.operators.prox_plus, pm.operators.prox_plus), ("user", my_plus, my_plus)):
        A, S = A0.copy(), S0.copy()
        ret = pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, step=step, max_iter=8, e_rel=1e-9)
        runs[name] = (A, S, ret)
    assert np.abs(runs["user"][0] - A0).max() > 1e-3, "the factors did not move"
    assert not all(runs["user"][2][0]), "8 iterations at e_rel = 1e-9 cannot have converged"
    np.testing.assert_array_equal(runs["user"][0], runs["lib"][0])
    np.testing.assert_array_equal(runs["user"][1], runs["lib"][1])
    np.testing.assert_allclose(np.array(runs["user"][2][2], dtype=np.float64), step.steps, rtol=1e-6)


def test_constant_step_with_user_prox_in_adaprox(pm, orc):
    """constant_step + a user-defined prox recomputed the default rule mean(X)/10 on the device and
    overwrote the constants.  constant_step + my_soft must match constant_step + the library's prox_soft."""
    Y, A0, S0 = orc.synthetic_problem(300, 500, 12, np.float32, seed=9)
    step = pm.nmf.constant_step(0.01, 0.02)
    out = []
    for pS in (partial(pm.operators.prox_soft, thresh=1e-3), my_soft):
        A, S = A0.copy(), S0.copy()
        ret = pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", step=step, max_iter=10, e_rel=1e-4, prox_S=pS)
        out.append((A, S, ret))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=2e-5, atol=2e-6)
    # and the oracle with the same constants
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("soft", 1e-3, "relative"), scheme="adam", max_iter=10, e_rel=1e-4,
                    step=lambda A_, S_, it: (np.float32(0.01), np.float32(0.02)))
    np.testing.assert_allclose(out[1][0], Ao, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out[1][1], So, rtol=2e-4, atol=2e-5)


def test_bsdmm_with_an_empty_update_order(pm, orc):
    """update_order=[] updates nothing for max_iter iterations (algorithms.py:800-846)."""
    Y, A0, S0 = orc.synthetic_problem(64, 96, 4, np.float32, seed=1)
    A, S = A0.copy(), S0.copy()
    seen = []
    conv = pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, update_order=[], max_iter=3, callback=lambda *X, it=None: seen.append(it))
    assert conv == [None, None] and seen == [0, 1, 2]
    np.testing.assert_array_equal(A, A0)
    np.testing.assert_array_equal(S, S0)


# ---- generic `grad` callables and user-defined bsdmm operators (host round trips) -----------
def _fixture_problem(z, meta, c, orc):
    tag = "unity" if c["unity_S"] else "plain"
    if "inputs_%s/Y" % tag in z.files:
        return z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
    return orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.dtype(meta["dtype"]).type, c["unity_S"], meta["seed"])


def _numpy_grad(Y):
    """a user-written gradient of 1/2 |A S - Y|^2: plain NumPy, no object of this library"""
    def grad(A, S):
        R = A @ S - Y
        return R @ S.T, A.T @ R
    return grad


@pytest.mark.parametrize("fname", ["nmf_33x47_k3_f64.npz", "nmf_200x1000_k5_f64.npz"])
def test_pgm_with_a_user_written_gradient_reproduces_the_fixture(pm, orc, fname, caplog):
    """algorithms.pgm(X, grad, step, prox) with ANY callable as `grad` (algorithms.py:12): the reference-generated fixture
    `pgm` must come out of `pgm([A, S], lambda A, S: <numpy>, step_pgm, prox=[prox_plus] * 2)`."""
    from test_gpu_nmf import assert_factors_close
    z, meta = load_golden(fname)
    Y, A0, S0 = _fixture_problem(z, meta, meta["cases"]["pgm"], orc)
    A, S = A0.copy(), S0.copy()
    with caplog.at_level(logging.WARNING, logger="proxmin"):
        conv, G, steps = pm.pgm([A, S], _numpy_grad(Y), pm.nmf.step_pgm, prox=[pm.operators.prox_plus] * 2,
                                max_iter=meta["max_iter"], e_rel=meta["e_rel"])
    assert_factors_close(A, z["pgm/A"], meta["dtype"], fname + " A")
    assert_factors_close(S, z["pgm/S"], meta["dtype"], fname + " S")
    # the returned gradient is the callable's own, at the last evaluation point
    assert G[0].shape == A.shape and G[1].shape == S.shape
    # accelerated + the library's damped rule: same numbers as the fused path with the device's own gradient
    A1, S1 = A0.copy(), S0.copy()
    pm.pgm([A1, S1], _numpy_grad(Y), pm.nmf.scaled_step_pgm(0.5), prox=[pm.operators.prox_plus] * 2, accelerated=True, max_iter=8, e_rel=1e-9)
    A2, S2 = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A2, S2, accelerated=True, step=pm.nmf.scaled_step_pgm(0.5), max_iter=8, e_rel=1e-9)
    np.testing.assert_allclose(A1, A2, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(S1, S2, rtol=2e-5, atol=2e-6)


def test_adaprox_with_a_user_written_gradient(pm, orc):
    """algorithms.adaprox with a NumPy gradient callable: the fixture `adam`, and the fused tail fed by the host gradient."""
    from test_gpu_nmf import assert_factors_close
    z, meta = load_golden("nmf_33x47_k3_f64.npz")
    Y, A0, S0 = _fixture_problem(z, meta, meta["cases"]["adam"], orc)
    A, S = A0.copy(), S0.copy()
    pm.adaprox([A, S], _numpy_grad(Y), pm.nmf.step_adaprox, prox=[pm.operators.prox_plus] * 2, scheme="adam",
               max_iter=meta["max_iter"], e_rel=meta["e_rel"])
    assert_factors_close(A, z["adam/A"], meta["dtype"], "adam A")
    assert_factors_close(S, z["adam/S"], meta["dtype"], "adam S")


def test_bsdmm_with_a_user_written_constraint_matches_the_fixture(pm, orc):
    """nmf(..., algorithm=bsdmm, proxs_g=[[prox_plus, <user soft threshold>]] * 2): the fixture `bsdmm_plus_soft`
    (generated from the reference with its own prox_soft); also a user-written prox_A / prox_S."""
    from test_gpu_nmf import assert_factors_close
    for fname in ("nmf_33x47_k3_f64.npz", "nmf_64x96_k8_f32.npz"):
        z, meta = load_golden(fname)
        c = meta["cases"]["bsdmm_plus_soft"]
        Y, A0, S0 = _fixture_problem(z, meta, c, orc)
        thresh = c["proxs_g"][0][1][1]
        user_soft = partial(my_soft, thresh=thresh)
        for pA, pS in ((pm.operators.prox_plus, pm.operators.prox_plus), (my_plus, my_plus)):
            A, S = A0.copy(), S0.copy()
            conv = pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.bsdmm, proxs_g=[[pm.operators.prox_plus, user_soft]] * 2,
                              max_iter=meta["max_iter"], e_rel=meta["e_rel"])
            assert_factors_close(A, z["bsdmm_plus_soft/A"], meta["dtype"], fname + " A")
            assert_factors_close(S, z["bsdmm_plus_soft/S"], meta["dtype"], fname + " S")
            assert len(conv) == 2


def test_bsdmm_with_generic_closures(pm, orc):
    """algorithms.bsdmm(X, proxs_f, steps_f_cb, proxs_g) with closures that are NOT the library's (algorithms.py:653): the
    reference's own idiom for NMF (nmf.py:181-193) written with NumPy; must match the tagged-closure (all-device) run."""
    Y, A0, S0 = orc.synthetic_problem(96, 140, 6, np.float32, seed=5)
    grad = _numpy_grad(Y)

    def prox_f(X, step, Xs=None, j=None):
        return np.maximum(X - step * grad(*Xs)[j], 0)

    def step_f(Xs, j=None):
        A, S = Xs
        L = np.linalg.eigvalsh(S @ S.T)[-1] if j == 0 else np.linalg.eigvalsh(A.T @ A)[-1]
        return 1.0 / L

    pg = [[pm.operators.prox_plus, partial(pm.operators.prox_soft, thresh=0.01)]] * 2
    A, S = A0.copy(), S0.copy()
    pm.bsdmm([A, S], prox_f, step_f, proxs_g=pg, max_iter=6, e_rel=1e-9)
    A2, S2 = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A2, S2, algorithm=pm.bsdmm, proxs_g=pg, max_iter=6, e_rel=1e-9)
    assert np.abs(A - A0).max() > 1e-3
    np.testing.assert_allclose(A, A2, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(S, S2, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_pgm_with_array_valued_user_steps(pm, tag):
    """A user `step` may return ARRAYS that broadcast against the blocks (algorithms.py:106-108 multiplies S[j] into G[j]
    and hands S[j] to prox[j]) -- round 3: uploaded element by element (pmx_pgm_step_arrays), the operators get their step
    per element too (prox_soft_plus with a relative threshold).  Fixture tests/golden/array_steps.npz, generated from the
    reference: per-component vectors, a full-shape array with a row array, a scalar next to a vector, plain and FISTA; a
    user-written prox is handed the array as it was returned; the arrays come back as pgm's third return value."""
    from test_gpu_nmf import assert_factors_close
    z, meta = load_golden("array_steps.npz")
    Y, A0, S0 = z[tag + "/Y"], z[tag + "/A0"], z[tag + "/S0"]
    vecA, vecS, fullA, rowS = (z["%s/%s" % (tag, k)] for k in ("vecA", "vecS", "fullA", "rowS"))
    dtype = "float64" if tag == "f64" else "float32"
    soft = partial(pm.operators.prox_soft_plus, thresh=0.05, type="relative")
    sA_scalar = float(vecA[0])           # = 0.5 * step_pgm's value for A, as the generator used
    runs = {
        "vectors_plus": ((vecA, vecS), [pm.operators.prox_plus] * 2, False),
        "vectors_plus_fista": ((vecA, vecS), [pm.operators.prox_plus] * 2, True),
        "full_row_soft": ((fullA, rowS), [soft, soft], False),
        "scalar_and_vector": ((sA_scalar, vecS), [pm.operators.prox_plus, soft], True),
    }
    assert sorted(runs) == meta["cases"][tag]["runs"]
    grad = partial(pm.nmf.grad_likelihood, Y=Y)
    for name, (st, prox, accel) in runs.items():
        A, S = A0.copy(), S0.copy()
        conv, G, steps = pm.pgm([A, S], grad, lambda *X, it=None, st=st: st, prox=prox, accelerated=accel, e_rel=1e-6, max_iter=10)
        assert_factors_close(A, z["%s/%s/A" % (tag, name)], dtype, "%s %s A" % (tag, name))
        assert_factors_close(S, z["%s/%s/S" % (tag, name)], dtype, "%s %s S" % (tag, name))
        for j in range(2):
            if np.ndim(st[j]):
                assert steps[j] is st[j]
            else:
                assert float(steps[j]) == pytest.approx(float(st[j]), rel=1e-6)
    # a user-written prox next to array steps: it receives the array itself (its shape says so), result as with the operator
    seen = []

    def my_plus_logged(X, step):
        seen.append(np.shape(step))
        return np.maximum(X, 0)
    A, S = A0.copy(), S0.copy()
    pm.pgm([A, S], grad, lambda *X, it=None: (vecA, vecS), prox=[my_plus_logged, pm.operators.prox_plus], e_rel=1e-6, max_iter=10)
    assert seen and all(s == vecA.shape for s in seen)
    assert_factors_close(A, z[tag + "/vectors_plus/A"], dtype, tag + " user prox A")
    assert_factors_close(S, z[tag + "/vectors_plus/S"], dtype, tag + " user prox S")
    # what NumPy would refuse, this refuses the same way
    with pytest.raises(ValueError):
        pm.pgm([A0.copy(), S0.copy()], grad, lambda *X, it=None: (np.ones(A0.shape[1] + 1), 1e-3), prox=[pm.operators.prox_plus] * 2, max_iter=2)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_backtracking_with_a_user_step(pm, tag):
    """algorithms.pgm(backtracking=True, f=...) with a user `step` (round 3): the callable on the host once per iteration,
    the Beck-Teboulle line search on the device (pmx_pgm_set_fixed_steps + pmx_pgm_run(ctx, 1)).  Fixture bt_user_step.npz,
    generated from the reference with 1.5 x / 2 x the Lipschitz steps (the search halves T) and, accelerated, 1 x."""
    from test_gpu_nmf import assert_factors_close
    z, meta = load_golden("bt_user_step.npz")
    Y, A0, S0 = z[tag + "/Y"], z[tag + "/A0"], z[tag + "/S0"]
    dtype = "float64" if tag == "f64" else "float32"
    grad = partial(pm.nmf.grad_likelihood, Y=Y)
    f = partial(pm.nmf.log_likelihood, Y=Y)
    seen = {"grads": 0}

    def scaled(fac):
        def st(*X, it=None):
            return tuple(fac * s for s in pm.nmf.step_pgm(*X))
        return st

    def step2_grads(*X, it=None, grads=None):
        seen["grads"] += int(grads is not None and grads[0].shape == X[0].shape)
        return tuple(2.0 * s for s in pm.nmf.step_pgm(*X))
    runs = {"x1.5": (scaled(1.5), False), "x2_with_grads": (step2_grads, False), "x1_fista": (scaled(1.0), True)}
    assert sorted(runs) == meta["cases"][tag]["runs"]
    for name, (st, accel) in runs.items():
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        conv, G, steps = pm.pgm([A, S], grad, st, prox=[pm.operators.prox_plus] * 2, accelerated=accel, backtracking=True, f=f,
                                e_rel=1e-6, max_iter=15, callback=tb)
        key = "%s/%s" % (tag, name)
        assert len(tb.trace) == int(z[key + "/n_callbacks"])
        assert_factors_close(A, z[key + "/A"], dtype, key + " A")
        assert_factors_close(S, z[key + "/S"], dtype, key + " S")
        assert pm.nmf.log_likelihood(A, S, Y=Y) == pytest.approx(float(z[key + "/loss"]), rel=2e-3)
    assert seen["grads"] >= 15
    # [r4] a user prox next to the line search runs too (test_backtracking_with_a_user_prox); what stays out: a user GRADIENT there
    with pytest.raises(NotImplementedError):
        pm.pgm([A0.copy(), S0.copy()], lambda A, S: pm.nmf.grad_likelihood(A, S, Y=Y), scaled(1.0), prox=[pm.operators.prox_plus] * 2,
               backtracking=True, f=f, max_iter=2)


@pytest.mark.parametrize("bbtype,accel", [(1, False), (2, True)])
def test_barzilai_borwein_steps_with_a_user_prox_or_gradient(pm, orc, bbtype, accel):
    """utils.BarzilaiBorweinStepper next to a user-written prox / gradient (round 3): the rule stays on the device -- it is the
    step, evaluated by the same kernels as in a fused iteration -- and the callable takes its host round trip.  A user-written
    projection and a NumPy gradient that compute what the library computes must give what the fused path gives."""
    Y, A0, S0 = orc.synthetic_problem(220, 310, 7, np.float32, seed=13)

    def run(prox, grad):
        A, S = A0.copy(), S0.copy()
        bb = pm.utils.BarzilaiBorweinStepper(type=bbtype, init_r=0.1)
        conv, G, steps = pm.pgm([A, S], grad, bb.step, prox=prox, accelerated=accel, max_iter=8, e_rel=1e-12)
        return A, S, steps
    lib_grad = partial(pm.nmf.grad_likelihood, Y=Y)
    A1, S1, st1 = run([pm.operators.prox_plus] * 2, lib_grad)                     # fused
    A2, S2, st2 = run([my_plus, pm.operators.prox_plus], lib_grad)                # block A's prox on the host
    assert np.array_equal(A1, A2) and np.array_equal(S1, S2)
    assert tuple(st1) == tuple(st2)
    A3, S3, st3 = run([pm.operators.prox_plus] * 2, _numpy_grad(Y))               # the gradient from the host (fp32 NumPy)
    np.testing.assert_allclose(A3, A1, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(S3, S1, rtol=2e-4, atol=2e-5)
    # and against the oracle's restatement of the stepper
    Ao, So = A0.copy(), S0.copy()
    obb = orc.BBStepper(kind=bbtype, init_r=0.1)
    orc.pgm_nmf(Y, Ao, So, step=lambda a, s, it, g: obb.step((a, s), it, g), accelerated=accel, max_iter=8, e_rel=1e-12)
    from test_gpu_nmf import assert_factors_close
    assert_factors_close(A2, Ao, np.float32, "bb + user prox A")
    assert_factors_close(S2, So, np.float32, "bb + user prox S")


@pytest.mark.parametrize("accel", [False, True])
def test_backtracking_with_a_user_prox(pm, orc, accel):
    """[r4] algorithms.py:110-127 with a user-written prox (round 3: NotImplementedError): every trial of that block takes a
    host round trip (pmx_pgm_bt_split), the sufficient-decrease test, the other block and f stay on the device.  A 4 x too
    long fixed step forces halvings.  A user-written projection must give the library operator's factors BIT FOR BIT (same
    device arithmetic on either side of an exact host operation); a user-written soft threshold is checked against the oracle;
    and the callable is handed T[j] S[j], the halved step, like the reference's (:125)."""
    from test_gpu_nmf import assert_factors_close
    Y, A0, S0 = orc.synthetic_problem(150, 210, 5, np.float32, seed=17)
    sA, sS = orc.lipschitz_steps(A0.astype(np.float64), S0.astype(np.float64))
    fixed = (4 * sA, 4 * sS)
    f = partial(pm.nmf.log_likelihood, Y=Y)
    seen = []

    def my_plus(X, step):
        seen.append(float(step))
        return np.maximum(X, 0)
    runs = {}
    for name, prox_A in (("library", pm.operators.prox_plus), ("user", my_plus)):
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        pm.nmf.nmf(Y, A, S, prox_A=prox_A, step=pm.nmf.constant_step(*fixed), accelerated=accel, backtracking=True, f=f,
                   max_iter=10, e_rel=1e-9, callback=tb)
        runs[name] = (A, S, len(tb.trace))
    np.testing.assert_array_equal(runs["user"][0], runs["library"][0])
    np.testing.assert_array_equal(runs["user"][1], runs["library"][1])
    assert runs["user"][2] == runs["library"][2]
    # the callable is handed T[j] S[j] (algorithms.py:108, :125): the fixed step times a power of 1/2
    assert seen[0] == pytest.approx(fixed[0], rel=1e-6)
    assert all(abs(np.log2(fixed[0] / s) - round(np.log2(fixed[0] / s))) < 1e-4 for s in seen), seen[:8]
    # both blocks with user callables, one of them a soft threshold, against the oracle
    seen_S = []

    def my_soft(X, step):
        seen_S.append(float(step))
        return np.sign(X) * np.maximum(np.abs(X) - 0.01 * step, 0)
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, prox_A=my_plus, prox_S=my_soft, step=pm.nmf.constant_step(*fixed), accelerated=accel, backtracking=True, f=f,
               max_iter=10, e_rel=1e-9)
    Ao, So = A0.copy(), S0.copy()
    orc.pgm_nmf(Y, Ao, So, prox_A=("plus",), prox_S=("soft", 0.01, "relative"), step=lambda a, s, it, g: fixed, accelerated=accel,
                backtracking=True, max_iter=10, e_rel=1e-9)
    assert_factors_close(A, Ao, np.float32, "user prox + backtracking A")
    assert_factors_close(S, So, np.float32, "user prox + backtracking S")
    assert min(seen + seen_S) < 0.6 * min(fixed), "a 4 x too long step must have been halved on one of the blocks"
    # a user `step` AND a user prox with the line search
    A, S = A0.copy(), S0.copy()
    pm.pgm([A, S], partial(pm.nmf.grad_likelihood, Y=Y), lambda *X, it=None: fixed, prox=[my_plus, pm.operators.prox_plus], accelerated=accel,
           backtracking=True, f=f, e_rel=1e-9, max_iter=10)
    np.testing.assert_array_equal(A, runs["library"][0])
    np.testing.assert_array_equal(S, runs["library"][1])


def test_barzilai_borwein_stepper_called_as_a_function(pm):
    """[r5] utils.BarzilaiBorweinStepper.step(*X, it=, grads=) on ndarrays, as the reference allows (utils.py:216-241): the recorded
    calls of the REAL reference (helpers.npz bb1 / bb2: six calls each on two fp64 blocks) -- same state, same return types, the
    six sums per block from the device (pmx_bb_sums).  Then the same calls on fp32 copies against the fp64 results."""
    from conftest import load_golden
    z, _ = load_golden("helpers.npz")
    for typ in (1, 2):
        for dt, rtol in ((np.float64, 1e-10), (np.float32, 2e-5)):
            bb = pm.utils.BarzilaiBorweinStepper(type=typ, init_r=0.1)
            for it in range(6):
                X = tuple(z["bb%d/X_%s_%d" % (typ, b, it)].astype(dt) for b in "AS")
                G = tuple(z["bb%d/G_%s_%d" % (typ, b, it)].astype(dt) for b in "AS")
                out = bb.step(*X, it=it, grads=G)
                assert isinstance(out, tuple) if it == 0 else isinstance(out, np.ndarray)
                np.testing.assert_allclose(np.asarray(out, dtype=np.float64), z["bb%d/steps" % typ][it], rtol=rtol, err_msg="type %d call %d %s" % (typ, it, dt.__name__))
            assert bb.X_[0].dtype == dt and bb.G_ is G
