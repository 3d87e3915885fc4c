"""GPU: fp64 arithmetic for fp64 inputs on the small-problem path (PMX_MODE_F64, proxmin_amd/csrc/k_small_f64.hip).

The reference keeps the dtype of its inputs (nmf.py:39-41) and its own examples -- and BASELINE cfg1, 200 x 1000 x 5 -- are
fp64.  Rounds 1-3 computed those in fp32 and cast back (1e-4-class agreement); the fp64 kernels are held to **rtol 1e-10**
against the reference's own fp64 fixtures here: gradient / likelihood at kernel level, the `pgm` and
`fista_half` rows of both fp64 fixture files end to end (factors, the first recorded iterates, the returned gradient and
steps, the converged flags), chained and with a per-iteration callback; and that everything the mode does not cover says so."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope="module")
def pm():
    import __graft_entry__ as g
    g.build()
    import proxmin_amd
    return proxmin_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


@pytest.mark.parametrize("M,N,K", [(200, 1000, 5), (33, 47, 3), (120, 500, 12), (4000, 200, 16), (8, 256, 1), (1024, 1024, 8)])
def test_fp64_gradient_and_likelihood(pm, orc, M, N, K):
    from proxmin_amd.engine import DeviceNMF
    Y, A, S = orc.synthetic_problem(M, N, K, np.float64, seed=M + N + K)
    with DeviceNMF(M, N, K, mode="f64") as dev:
        assert dev.k1_info()["kernel"] == "k64_front"
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
        loss = dev.loglike()
        A2, S2 = dev.get_factors()
    assert gA.dtype == np.float64 and gS.dtype == np.float64
    np.testing.assert_array_equal(A2, A)
    np.testing.assert_array_equal(S2, S)
    rA, rS = orc.residual_gradients(A, S, Y)
    np.testing.assert_allclose(gA, rA, rtol=1e-12, atol=1e-12 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=1e-12, atol=1e-12 * np.abs(rS).max())
    assert loss == pytest.approx(orc.half_sq_residual(A, S, Y), rel=1e-13)


@pytest.mark.parametrize("with_callback", [False, True])
@pytest.mark.parametrize("fname", ["nmf_200x1000_k5_f64.npz", "nmf_33x47_k3_f64.npz"])
def test_fp64_fixtures_pgm_rows_to_ten_digits(pm, orc, monkeypatch, fname, with_callback):
    from test_gpu_nmf import run_device_case
    from proxmin_amd import algorithms, engine
    modes = []
    real = engine.DeviceNMF

    class Spy(real):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            modes.append(self.mode)
    monkeypatch.setattr(algorithms, "DeviceNMF", Spy)
    z, meta = load_golden(fname)
    assert meta["dtype"] == "float64"
    done = 0
    for name, c in meta["cases"].items():
        if c["alg"] != "pgm":
            continue
        tag = "unity" if c["unity_S"] else "plain"
        if "inputs_%s/Y" % tag in z.files:
            Y, A0, S0 = z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
        else:
            Y, A0, S0 = orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.float64, c["unity_S"], meta["seed"])
        assert Y.dtype == np.float64 and A0.dtype == np.float64
        tb = pm.utils.Traceback() if with_callback else None
        del modes[:]
        A, S, ret = run_device_case(pm, c, Y, A0, S0, meta["max_iter"], meta["e_rel"], callback=tb)
        assert modes == ["f64"], modes                       # the fp64 kernels are what ran
        assert A.dtype == np.float64
        np.testing.assert_allclose(A, z[name + "/A"], rtol=RTOL, atol=1e-14, err_msg="%s %s A" % (fname, name))
        np.testing.assert_allclose(S, z[name + "/S"], rtol=RTOL, atol=1e-14, err_msg="%s %s S" % (fname, name))
        conv, G, steps = ret
        np.testing.assert_allclose(G[0], z[name + "/G_A"], rtol=1e-8, atol=1e-10 * float(np.abs(z[name + "/G_A"]).max()))
        np.testing.assert_allclose(G[1], z[name + "/G_S"], rtol=1e-8, atol=1e-10 * float(np.abs(z[name + "/G_S"]).max()))
        np.testing.assert_allclose(np.array(steps, dtype=np.float64), z[name + "/steps"], rtol=1e-10)
        assert list(conv) == list(z[name + "/conv"])
        if with_callback:
            assert len(tb.trace) == int(z[name + "/n_callbacks"]), name
            i = 0
            while "%s/trace_A_%d" % (name, i) in z.files:
                np.testing.assert_allclose(tb.trace[i][0], z["%s/trace_A_%d" % (name, i)], rtol=RTOL, atol=1e-14)
                np.testing.assert_allclose(tb.trace[i][1], z["%s/trace_S_%d" % (name, i)], rtol=RTOL, atol=1e-14)
                i += 1
            assert i >= 3 or ("%s/trace_A_0" % name) not in z.files      # (the large fixture records no iterates)
        done += 1
    assert done >= 2


@pytest.mark.parametrize("with_callback", [False, True])
@pytest.mark.parametrize("fname", ["nmf_200x1000_k5_f64.npz", "nmf_33x47_k3_f64.npz"])
def test_fp64_fixtures_adaprox_rows_to_ten_digits(pm, orc, monkeypatch, fname, with_callback):
    """[r4] every adaprox row of the reference's fp64 fixtures (six moment schemes, prox_unity_plus on either block, a relative soft
    threshold, prox=None) on the fp64 kernels (k64_front + k64_ada_iter): factors, recorded iterates, first and second moments to
    rtol 1e-10; RAdam while the reference's own iterates are finite."""
    from test_gpu_nmf import run_device_case
    from proxmin_amd import algorithms, engine
    modes = []
    real = engine.DeviceNMF

    class Spy(real):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            modes.append(self.mode)
    monkeypatch.setattr(algorithms, "DeviceNMF", Spy)
    z, meta = load_golden(fname)
    assert meta["dtype"] == "float64"
    done = 0
    for name, c in meta["cases"].items():
        if c["alg"] != "adaprox":
            continue
        tag = "unity" if c["unity_S"] else "plain"
        if "inputs_%s/Y" % tag in z.files:
            Y, A0, S0 = z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
        else:
            Y, A0, S0 = orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.float64, c["unity_S"], meta["seed"])
        finite = np.isfinite(z[name + "/A"]).all() and np.isfinite(z[name + "/S"]).all() and max(np.abs(z[name + "/A"]).max(), np.abs(z[name + "/S"]).max()) < 1e30
        tb = pm.utils.Traceback() if with_callback else None
        del modes[:]
        A, S, ret = run_device_case(pm, c, Y, A0, S0, meta["max_iter"], meta["e_rel"], callback=tb)
        assert modes == ["f64"], (name, modes)               # the fp64 kernels are what ran
        assert A.dtype == np.float64
        conv, Mm, Vv, Vh = ret
        if finite:
            for got, key in ((A, "A"), (S, "S"), (Mm[0], "M_A"), (Mm[1], "M_S"), (Vv[0], "V_A"), (Vv[1], "V_S")):
                want = z[name + "/" + key]
                np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-13 * max(1.0, float(np.abs(want).max())), err_msg="%s %s %s" % (fname, name, key))
            assert list(conv) == list(z[name + "/conv"])
        if with_callback:
            assert len(tb.trace) == int(z[name + "/n_callbacks"]), name
            i = 0
            while "%s/trace_A_%d" % (name, i) in z.files:
                rA, rS = z["%s/trace_A_%d" % (name, i)], z["%s/trace_S_%d" % (name, i)]
                if not (np.isfinite(rA).all() and np.isfinite(rS).all()) or max(np.abs(rA).max(), np.abs(rS).max()) > 1e30:
                    break
                np.testing.assert_allclose(tb.trace[i][0], rA, rtol=RTOL, atol=1e-13 * max(1.0, float(np.abs(rA).max())), err_msg="%s iterate %d" % (name, i))
                np.testing.assert_allclose(tb.trace[i][1], rS, rtol=RTOL, atol=1e-13 * max(1.0, float(np.abs(rS).max())), err_msg="%s iterate %d" % (name, i))
                i += 1
        done += 1
    assert done >= 3


def test_fp64_adaprox_warm_start_fixed_steps_and_convergence(pm, orc):
    """warm start with M / V / Vhat arrays (the true AMSGrad running maximum, algorithms.py:356-359), array-valued b1, two constant
    steps (nmf.constant_step), and a run that converges: against the fp64 oracle to 1e-10, converged flags and pass counts equal"""
    from functools import partial
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(150, 333, 6, np.float64, unity_S=True, seed=9)
    rng = np.random.default_rng(3)
    # (1) warm start, amsgrad with Vhat, b1 array
    b1 = np.linspace(0.9, 0.5, 9)
    M0 = [rng.normal(size=A0.shape) * 0.01, rng.normal(size=S0.shape) * 0.01]
    V0 = [rng.random(A0.shape) * 1e-3, rng.random(S0.shape) * 1e-3]
    Vh0 = [v * 1.5 for v in V0]
    A, S = A0.copy(), S0.copy()
    M, V, Vh = [m.copy() for m in M0], [v.copy() for v in V0], [v.copy() for v in Vh0]
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(ops.prox_unity_plus, axis=0), b1=b1, max_iter=9, e_rel=1e-4,
               M=M, V=V, Vhat=Vh, check_convergence=False)
    Ao, So = A0.copy(), S0.copy()
    Mo, Vo, Vho = [m.copy() for m in M0], [v.copy() for v in V0], [v.copy() for v in Vh0]
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0), scheme="amsgrad", b1=b1, max_iter=9, e_rel=1e-4, M=Mo, V=Vo, Vhat=Vho, check_convergence=False)
    for got, want in ((A, Ao), (S, So), (M[0], Mo[0]), (V[1], Vo[1]), (Vh[0], Vho[0]), (Vh[1], Vho[1])):
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-14)
    # (2) constant steps
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", step=pm.nmf.constant_step(0.01, 0.002), max_iter=7, e_rel=1e-4, check_convergence=False)
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, step=lambda a, s, it: (0.01, 0.002), scheme="adam", max_iter=7, e_rel=1e-4, check_convergence=False)
    np.testing.assert_allclose(A, Ao, rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(S, So, rtol=RTOL, atol=1e-14)
    # (3) a run that converges: same iteration count, flags, factors
    A, S = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv, _, _, _ = pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", max_iter=400, e_rel=2e-3, callback=tb)
    Ao, So = A0.copy(), S0.copy()
    oret = orc.adaprox_nmf(Y, Ao, So, scheme="adam", max_iter=400, e_rel=2e-3)
    assert tuple(conv) == tuple(oret[0]) and len(tb.trace) == oret[4]
    np.testing.assert_allclose(A, Ao, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(S, So, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("with_callback", [False, True])
@pytest.mark.parametrize("fname", ["nmf_200x1000_k5_f64.npz", "nmf_33x47_k3_f64.npz"])
def test_fp64_fixtures_bsdmm_rows_to_ten_digits(pm, orc, monkeypatch, fname, with_callback):
    """[r4] the bsdmm rows of the reference's fp64 fixtures (no constraints, plus + soft on both blocks, a constraint on one
    block only) on the fp64 kernels (k64_front + k64_bsdmm_block): factors, recorded iterates and flags; Z / U against the oracle"""
    from test_gpu_nmf import run_device_case
    from proxmin_amd import algorithms, engine
    modes = []
    real = engine.DeviceNMF

    class Spy(real):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            modes.append(self.mode)
    monkeypatch.setattr(algorithms, "DeviceNMF", Spy)
    z, meta = load_golden(fname)
    done = 0
    for name, c in meta["cases"].items():
        if c["alg"] != "bsdmm":
            continue
        tag = "unity" if c["unity_S"] else "plain"
        if "inputs_%s/Y" % tag in z.files:
            Y, A0, S0 = z["inputs_%s/Y" % tag], z["inputs_%s/A0" % tag], z["inputs_%s/S0" % tag]
        else:
            Y, A0, S0 = orc.synthetic_problem(meta["M"], meta["N"], meta["K"], np.float64, c["unity_S"], meta["seed"])
        tb = pm.utils.Traceback() if with_callback else None
        del modes[:]
        A, S, ret = run_device_case(pm, c, Y, A0, S0, meta["max_iter"], meta["e_rel"], callback=tb)
        assert modes == ["f64"], (name, modes)
        np.testing.assert_allclose(A, z[name + "/A"], rtol=RTOL, atol=1e-14, err_msg="%s %s A" % (fname, name))
        np.testing.assert_allclose(S, z[name + "/S"], rtol=RTOL, atol=1e-14, err_msg="%s %s S" % (fname, name))
        assert [bool(x) for x in ret] == [bool(x) for x in z[name + "/conv"]]
        if with_callback:
            assert len(tb.trace) == int(z[name + "/n_callbacks"]), name
            i = 0
            while "%s/trace_A_%d" % (name, i) in z.files:
                np.testing.assert_allclose(tb.trace[i][0], z["%s/trace_A_%d" % (name, i)], rtol=RTOL, atol=1e-14)
                np.testing.assert_allclose(tb.trace[i][1], z["%s/trace_S_%d" % (name, i)], rtol=RTOL, atol=1e-14)
                i += 1
        done += 1
    assert done >= 2


def test_fp64_bsdmm_constraint_variables_and_convergence(pm, orc):
    """Z_i / U_i of both blocks against the oracle's (the reference drops them), and a run that stops by Boyd's test at the
    oracle's iteration"""
    from functools import partial
    from proxmin_amd import _lib
    from proxmin_amd.engine import DeviceNMF
    ops = pm.operators
    M, N, K = 90, 210, 7
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, seed=5)
    pA, pS = ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(ops.prox_plus, 1)
    gA = [ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_soft, thresh=0.02), 0)]
    gS = [ops.device_proxseq(partial(ops.prox_soft_plus, thresh=0.01), 1)]
    with DeviceNMF(M, N, K, mode="f64") as dev:
        dev.set_Y(Y)
        dev.set_factors(A0, S0)
        dev.bsdmm_begin([pA, pS], [gA, gS], e_rel=(1e-9, 1e-9), e_abs=(0.0, 0.0))
        dev.bsdmm_run(8)
        A, S = dev.get_factors()
        Z = [[dev._download(_lib.BUF_Z0 + j * _lib.MAX_G + i, (M, N)[j]) for i in range((2, 1)[j])] for j in range(2)]
        U = [[dev._download(_lib.BUF_U0 + j * _lib.MAX_G + i, (M, N)[j]) for i in range((2, 1)[j])] for j in range(2)]
    Ao, So = A0.copy(), S0.copy()
    state = {}
    orc.bsdmm_nmf(Y, Ao, So, proxs_g=[[("plus",), ("soft", 0.02, "relative")], [("soft_plus", 0.01, "relative")]], max_iter=8, e_rel=1e-9, state=state)
    np.testing.assert_allclose(A, Ao, rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(S, So, rtol=RTOL, atol=1e-14)
    for i in range(2):
        np.testing.assert_allclose(Z[0][i], state["Z"][0][i], rtol=RTOL, atol=1e-14)
        np.testing.assert_allclose(U[0][i], state["U"][0][i], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(Z[1][0].T, state["Z"][1][0], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(U[1][0].T, state["U"][1][0], rtol=1e-8, atol=1e-13)
    # convergence by Boyd's test
    A, S = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv = pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, max_iter=500, e_rel=5e-3, callback=tb)
    Ao, So = A0.copy(), S0.copy()
    oconv, oit = orc.bsdmm_nmf(Y, Ao, So, max_iter=500, e_rel=5e-3)
    assert list(conv) == list(oconv) and len(tb.trace) == oit
    np.testing.assert_allclose(A, Ao, rtol=1e-8, atol=1e-12)


def test_fp64_operators_and_convergence(pm, orc):
    """every device operator in fp64 inside pgm (unity along the short axis, soft threshold, alternating projections) and a run
    that converges: iteration count and flags equal to the oracle's, factors to 1e-10"""
    from functools import partial
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(150, 333, 6, np.float64, unity_S=True, seed=9)
    cases = [
        (dict(prox_S=partial(ops.prox_unity_plus, axis=0)), dict(prox_S=("unity_plus", 0)), 12, 1e-9),
        (dict(prox_A=partial(ops.prox_soft, thresh=0.01), prox_S=partial(ops.prox_hard, thresh=1e-3, type="absolute")),
         dict(prox_A=("soft", 0.01, "relative"), prox_S=("hard", 1e-3, "absolute")), 12, 1e-9),
        (dict(), dict(), 400, 2e-2),                         # converges: the stopping test stops both at the same iteration
    ]
    for kw, okw, its, e_rel in cases:
        A, S = A0.copy(), S0.copy()
        conv, G, steps = pm.nmf.nmf(Y, A, S, max_iter=its, e_rel=e_rel, **kw)
        Ao, So = A0.copy(), S0.copy()
        oret = orc.pgm_nmf(Y, Ao, So, max_iter=its, e_rel=e_rel, **okw)
        np.testing.assert_allclose(A, Ao, rtol=RTOL, atol=1e-14)
        np.testing.assert_allclose(S, So, rtol=RTOL, atol=1e-14)
        assert tuple(conv) == tuple(oret[0])


def test_fp64_mode_says_what_it_does_not_cover(pm, orc, monkeypatch):
    from proxmin_amd import _lib
    from proxmin_amd.engine import DeviceNMF
    with monkeypatch.context() as mp:                        # [r6] larger problems have their own fp64 kernels (tests/test_gpu_f64_big.py); without them:
        mp.setenv("PMX_F64_BIG", "0")
        with pytest.raises(NotImplementedError):             # not a small problem (PMX_E_UNSUPPORTED)
            DeviceNMF(4096, 4096, 32, mode="f64")
    Y, A, S = orc.synthetic_problem(64, 96, 4, np.float64, seed=1)
    with DeviceNMF(64, 96, 4, mode="f64") as dev:
        dev.set_Y(Y)
        dev.set_factors(A, S)
        with pytest.raises(NotImplementedError):             # (the pieces for user callables exist in fp32 only)
            dev.pgm_begin([pm.operators.device_proxseq(pm.operators.prox_plus, j) for j in range(2)], backtracking=True, e_rel=(1e-3, 1e-3))
        with pytest.raises(NotImplementedError):
            dev.step_adaprox()
        sA, sS = dev.step_pgm()                              # nmf.step_pgm (nmf.py:44-65) to fp64 round-off
        LA, LS = orc.lipschitz_steps(A, S)
        assert sA == pytest.approx(LA, rel=1e-12) and sS == pytest.approx(LS, rel=1e-12)
    # fp64 inputs outside the mode's coverage still run (in fp32, cast back): a user-written prox on a small problem, pgm on a large one
    A1, S1 = A.copy(), S.copy()
    pm.nmf.nmf(Y, A1, S1, prox_A=lambda X, step: np.maximum(X, 0), max_iter=3, e_rel=1e-3)
    assert A1.dtype == np.float64 and np.isfinite(A1).all()
