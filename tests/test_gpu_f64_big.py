"""GPU: fp64 arithmetic for fp64 inputs at ANY size up to K = 128 (PMX_MODE_F64 outside the small-problem kernels:
proxmin_amd/csrc/k_big_f64.hip -- one v_mfma_f64_16x16x4_f64 pass per gradient, the step rule from fp64 Gram matrices, the three
back-ends' updates as plain launches).

The reference keeps the dtype of its inputs (nmf.py:39-41); until round 6 fp64 callers above K = 16 / M N = 2^20 were computed in
fp32 and cast back.  Everything here is held against the fp64 oracle: gradient, likelihood and step rule to round-off, the three
back-ends end to end to rtol 1e-9 (the sums run in another order than NumPy's; nothing else differs), stopping iterations and
proximal pass counts equal."""
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-9

SHAPES = [(384, 512, 64), (1000, 1500, 50), (256, 300, 128), (1500, 2000, 5), (65, 70, 17), (2048, 1024, 32), (130, 9000, 100), (64, 64, 33),
          (3, 400000, 2), (20000, 7, 3), (1, 70, 128), (129, 63, 65), (4097, 300, 1)]


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


def _close(got, want, rtol=RTOL, name=""):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * np.abs(want).max(), err_msg=name)


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gradient_likelihood_and_step_rule(pm, orc, M, N, K):
    from proxmin_amd.engine import DeviceNMF, f64_applies
    assert f64_applies(M, N, K)
    Y, A, S = orc.synthetic_problem(M, N, K, np.float64, seed=M + N + K)
    with DeviceNMF(M, N, K, mode="f64") as dev:
        assert dev.k1_info()["kernel"] == "k64_grad_pass"
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
        loss = dev.loglike()
        sA, sS = dev.step_pgm()
        A2, S2 = dev.get_factors()
    assert gA.dtype == np.float64 and gS.dtype == np.float64
    np.testing.assert_array_equal(A2, A)
    np.testing.assert_array_equal(S2, S)
    rA, rS = orc.residual_gradients(A, S, Y)
    np.testing.assert_allclose(gA, rA, rtol=1e-12, atol=1e-12 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=1e-12, atol=1e-12 * np.abs(rS).max())
    assert loss == pytest.approx(orc.half_sq_residual(A, S, Y), rel=1e-12)
    LA, LS = orc.lipschitz_steps(A, S)
    assert sA == pytest.approx(LA, rel=1e-11) and sS == pytest.approx(LS, rel=1e-11)


def test_gradient_of_an_asymmetric_problem_catches_a_transposed_tile(pm, orc):
    """Y with a structure no transposition leaves alone (rows and columns scaled differently, K components of distinct size)"""
    from proxmin_amd.engine import DeviceNMF
    M, N, K = 200, 333, 40
    rng = np.random.default_rng(2)
    A = rng.random((M, K)) * np.arange(1, K + 1)[None, :]
    S = rng.random((K, N)) * np.linspace(0.1, 3.0, N)[None, :]
    Y = rng.random((M, N)) * np.arange(1, M + 1)[:, None]
    with DeviceNMF(M, N, K, mode="f64") as dev:
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
    rA, rS = orc.residual_gradients(A, S, Y)
    np.testing.assert_allclose(gA, rA, rtol=1e-12, atol=1e-12 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=1e-12, atol=1e-12 * np.abs(rS).max())


def _spy(monkeypatch):
    from proxmin_amd import algorithms, engine
    seen = []
    real = engine.DeviceNMF

    class Spy(real):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            seen.append((self.mode, self.k1_info()["kernel"]))
    monkeypatch.setattr(algorithms, "DeviceNMF", Spy)
    return seen


@pytest.mark.parametrize("M,N,K", [(384, 512, 64), (1000, 1500, 50), (256, 300, 128), (1500, 2000, 5)])
@pytest.mark.parametrize("accelerated", [False, True])
def test_pgm_and_fista(pm, orc, monkeypatch, M, N, K, accelerated):
    seen = _spy(monkeypatch)
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=True, seed=7)
    kw = dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)) if accelerated else {}
    step = (lambda A_, S_, it, grads: tuple(0.5 * s_ for s_ in orc.lipschitz_steps(A_, S_))) if accelerated else None
    A, S = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv, G, steps = pm.nmf.nmf(Y, A, S, prox_S=partial(ops.prox_unity_plus, axis=0), max_iter=10, e_rel=1e-9, callback=tb, **kw)
    assert seen == [("f64", "k64_grad_pass")], seen
    Ac, Sc = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, Ac, Sc, prox_S=partial(ops.prox_unity_plus, axis=0), max_iter=10, e_rel=1e-9, **kw)      # chained: the same bits
    assert np.array_equal(A, Ac) and np.array_equal(S, Sc)
    Ao, So = A0.copy(), S0.copy()
    trace = []
    oret = orc.pgm_nmf(Y, Ao, So, prox_S=("unity_plus", 0), max_iter=10, e_rel=1e-9, accelerated=accelerated, step=step, trace=trace)
    assert A.dtype == np.float64
    _close(A, Ao, name="A")
    _close(S, So, name="S")
    assert tuple(conv) == tuple(oret[0])
    _close(tb.trace[3][0], trace[3][0], name="iterate 3")
    _close(G[0], oret[1][0], rtol=1e-8, name="returned gradient")


def test_pgm_operators_and_stopping_iteration(pm, orc):
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(300, 700, 24, np.float64, seed=9)
    cases = [
        (dict(prox_A=partial(ops.prox_soft, thresh=0.01), prox_S=partial(ops.prox_hard, thresh=1e-3, type="absolute")),
         dict(prox_A=("soft", 0.01, "relative"), prox_S=("hard", 1e-3, "absolute")), 12, 1e-9),
        (dict(prox_A=partial(ops.prox_unity, axis=1), prox_S=partial(ops.prox_soft_plus, thresh=0.02)),
         dict(prox_A=("unity", 1), prox_S=("soft_plus", 0.02, "relative")), 8, 1e-9),
        (dict(), dict(), 400, 2e-2),                         # converges: the stopping test stops both at the same iteration
    ]
    for kw, okw, its, e_rel in cases:
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        conv, G, steps = pm.nmf.nmf(Y, A, S, max_iter=its, e_rel=e_rel, callback=tb, **kw)
        Ao, So = A0.copy(), S0.copy()
        trace = []
        oret = orc.pgm_nmf(Y, Ao, So, max_iter=its, e_rel=e_rel, trace=trace, **okw)
        _close(A, Ao)
        _close(S, So)
        assert tuple(conv) == tuple(oret[0]) and len(tb.trace) == len(trace)
    assert len(trace) < 400


@pytest.mark.parametrize("scheme", ["adam", "nadam", "amsgrad", "padam", "adamx", "radam"])
def test_adaprox_schemes(pm, orc, monkeypatch, scheme):
    seen = _spy(monkeypatch)
    ops = pm.operators
    M, N, K = 320, 448, 48
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=True, seed=3)
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme=scheme, prox_S=partial(ops.prox_unity_plus, axis=0), max_iter=8, e_rel=1e-4, check_convergence=False)
    assert seen == [("f64", "k64_grad_pass")], seen
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0), scheme=scheme, max_iter=8, e_rel=1e-4, check_convergence=False)
    _close(A, Ao, name=scheme + " A")
    _close(S, So, name=scheme + " S")


@pytest.mark.parametrize("M,N,K", [(1000, 1500, 50), (256, 300, 128), (1500, 2000, 5)])
def test_adaprox_shapes_pass_counts_warm_start_and_constant_steps(pm, orc, M, N, K):
    from proxmin_amd.engine import DeviceNMF
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=True, seed=11)
    # (1) the proximal loops end after the oracle's number of passes (a tight e_rel: more passes than the first guess -- the chain is
    #     halted in front of the verdict and resumed, pmx_api.hip: HALT_NEED_SUB)
    its = 6
    with DeviceNMF(M, N, K, mode="f64") as dev:
        dev.set_Y(Y)
        dev.set_factors(A0, S0)
        dev.adaprox_begin([ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_unity_plus, axis=0), 1)],
                          scheme="amsgrad", check_convergence=False, prox_max_iter=1000, e_rel=(1e-7, 1e-7))
        res = dev.adaprox_run(np.full(its, 0.9), 0.9)
        got = [int(res.sub_iterations[0]), int(res.sub_iterations[1])]
        A, S = dev.get_factors()
    Ao, So = A0.copy(), S0.copy()
    out = orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=its, e_rel=1e-7, check_convergence=False)
    assert res.iterations == its and got == [int(out[5][0]), int(out[5][1])], (got, out[5])
    assert got[1] > 4 * its, "the case is meant to need more passes than the first guess (%r)" % (got,)
    _close(A, Ao)
    _close(S, So)
    # (2) prox_max_iter cuts the loop short
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", prox_S=partial(ops.prox_unity_plus, axis=0), max_iter=4, e_rel=1e-7, prox_max_iter=3, check_convergence=False)
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0), scheme="adam", max_iter=4, e_rel=1e-7, prox_max_iter=3, check_convergence=False)
    _close(A, Ao)
    _close(S, So)
    # (3) warm start with M / V / Vhat, b1 array
    rng = np.random.default_rng(3)
    b1 = np.linspace(0.9, 0.5, 7)
    M0 = [rng.normal(size=A0.shape) * 0.01, rng.normal(size=S0.shape) * 0.01]
    V0 = [rng.random(A0.shape) * 1e-3, rng.random(S0.shape) * 1e-3]
    Vh0 = [v * 1.5 for v in V0]
    A, S = A0.copy(), S0.copy()
    Mm, V, Vh = [m.copy() for m in M0], [v.copy() for v in V0], [v.copy() for v in Vh0]
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adamx", prox_S=partial(ops.prox_unity_plus, axis=0), b1=b1, max_iter=7, e_rel=1e-4,
               M=Mm, V=V, Vhat=Vh, check_convergence=False)
    Ao, So = A0.copy(), S0.copy()
    Mo, Vo, Vho = [m.copy() for m in M0], [v.copy() for v in V0], [v.copy() for v in Vh0]
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0), scheme="adamx", b1=b1, max_iter=7, e_rel=1e-4, M=Mo, V=Vo, Vhat=Vho, check_convergence=False)
    for got_, want in ((A, Ao), (S, So), (Mm[0], Mo[0]), (V[1], Vo[1]), (Vh[0], Vho[0]), (Vh[1], Vho[1])):
        _close(got_, want)
    # (4) constant steps, no prox on A
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", step=pm.nmf.constant_step(0.01, 0.002), max_iter=5, e_rel=1e-4, check_convergence=False)
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, step=lambda a, s, it: (0.01, 0.002), scheme="adam", max_iter=5, e_rel=1e-4, check_convergence=False)
    _close(A, Ao)
    _close(S, So)


def test_adaprox_outer_convergence(pm, orc):
    Y, A0, S0 = orc.synthetic_problem(300, 500, 20, np.float64, seed=9)
    A, S = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv, _, _, _ = pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="adam", max_iter=400, e_rel=2e-3, callback=tb)
    Ao, So = A0.copy(), S0.copy()
    oret = orc.adaprox_nmf(Y, Ao, So, scheme="adam", max_iter=400, e_rel=2e-3)
    assert tuple(conv) == tuple(oret[0]) and len(tb.trace) == oret[4] and oret[4] < 400
    _close(A, Ao, rtol=1e-8)
    _close(S, So, rtol=1e-8)
    # chained: the same iteration count and bits as one iteration per call
    Ac, Sc = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, Ac, Sc, algorithm=pm.adaprox, scheme="adam", max_iter=400, e_rel=2e-3)
    assert np.array_equal(A, Ac) and np.array_equal(S, Sc)


@pytest.mark.parametrize("M,N,K", [(384, 512, 64), (1000, 1500, 50), (256, 300, 128), (1500, 2000, 5)])
def test_bsdmm(pm, orc, monkeypatch, M, N, K):
    from proxmin_amd import _lib
    from proxmin_amd.engine import DeviceNMF
    seen = _spy(monkeypatch)
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, seed=5)
    # through nmf(): plus + soft on both blocks
    A, S = A0.copy(), S0.copy()
    conv = pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2, max_iter=8, e_rel=1e-9)
    assert seen == [("f64", "k64_grad_pass")], seen
    Ao, So = A0.copy(), S0.copy()
    oconv, oit = orc.bsdmm_nmf(Y, Ao, So, proxs_g=[[("plus",), ("soft", 1e-3, "relative")]] * 2, max_iter=8, e_rel=1e-9)
    _close(A, Ao)
    _close(S, So)
    assert list(conv) == list(oconv)
    # the constraint variables against the oracle's (the reference drops them)
    pA, pS = ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(ops.prox_plus, 1)
    gA = [ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_soft, thresh=0.02), 0)]
    gS = [ops.device_proxseq(partial(ops.prox_soft_plus, thresh=0.01), 1)]
    with DeviceNMF(M, N, K, mode="f64") as dev:
        dev.set_Y(Y)
        dev.set_factors(A0, S0)
        dev.bsdmm_begin([pA, pS], [gA, gS], e_rel=(1e-9, 1e-9), e_abs=(0.0, 0.0))
        dev.bsdmm_run(6)
        A, S = dev.get_factors()
        Z = [[dev._download(_lib.BUF_Z0 + j * _lib.MAX_G + i, (M, N)[j]) for i in range((2, 1)[j])] for j in range(2)]
        U = [[dev._download(_lib.BUF_U0 + j * _lib.MAX_G + i, (M, N)[j]) for i in range((2, 1)[j])] for j in range(2)]
    Ao, So = A0.copy(), S0.copy()
    state = {}
    orc.bsdmm_nmf(Y, Ao, So, proxs_g=[[("plus",), ("soft", 0.02, "relative")], [("soft_plus", 0.01, "relative")]], max_iter=6, e_rel=1e-9, state=state)
    _close(A, Ao)
    _close(S, So)
    for i in range(2):
        _close(Z[0][i], state["Z"][0][i])
        _close(U[0][i], state["U"][0][i], rtol=1e-7)
    _close(Z[1][0].T, state["Z"][1][0])
    _close(U[1][0].T, state["U"][1][0], rtol=1e-7)


def test_bsdmm_stops_by_boyds_test_at_the_oracles_iteration(pm, orc):
    """(this noisy problem meets Boyd's criteria only with an absolute tolerance: e_abs = 1 stops the oracle at iteration 232)"""
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(200, 400, 24, np.float64, seed=5)
    A, S = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv = pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus], [ops.prox_plus]], max_iter=400, e_rel=1e-3, e_abs=1.0, callback=tb)
    Ao, So = A0.copy(), S0.copy()
    oconv, oit = orc.bsdmm_nmf(Y, Ao, So, proxs_g=[[("plus",)], [("plus",)]], max_iter=400, e_rel=1e-3, e_abs=1.0)
    assert list(conv) == list(oconv) == [True, True] and len(tb.trace) == oit and oit < 400
    _close(A, Ao, rtol=1e-8)
    _close(S, So, rtol=1e-8)


@pytest.mark.parametrize("M,N,K", [(384, 512, 64), (1000, 1500, 50), (256, 300, 128), (90, 210, 7), (65, 70, 17)])
def test_weighted_likelihood(pm, orc, monkeypatch, M, N, K):
    """nmf.py:13-41 with an M x N weight array, fp64: D = W (A S - Y), 1/2 sum W (A S - Y)^2 -- kernel level, then adaprox and pgm (with an
    explicit step: the reference's default rule raises on a weight array, nmf.py:63) end to end.  Small shapes run the matrix-core kernels as
    well (mode "f64mfma": the fused small-problem kernels take no weights)."""
    from proxmin_amd.engine import DeviceNMF
    ops = pm.operators
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=True, seed=M + K)
    rng = np.random.default_rng(17)
    W = 0.2 + rng.random((M, N))
    W[rng.random((M, N)) < 0.1] = 0
    with DeviceNMF(M, N, K, mode="f64mfma") as dev:
        assert dev.k1_info()["kernel"] == "k64_grad_pass"
        dev.set_Y(Y)
        dev.set_W(W)
        dev.set_factors(A0, S0)
        gA, gS = dev.grad()
        loss = dev.loglike()
        dev.set_W(None)
        gA1, gS1 = dev.grad()
    rA, rS = orc.residual_gradients(A0, S0, Y, W)
    np.testing.assert_allclose(gA, rA, rtol=1e-12, atol=1e-12 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=1e-12, atol=1e-12 * np.abs(rS).max())
    assert loss == pytest.approx(orc.half_sq_residual(A0, S0, Y, W), rel=1e-12)
    uA, uS = orc.residual_gradients(A0, S0, Y)
    np.testing.assert_allclose(gA1, uA, rtol=1e-12, atol=1e-12 * np.abs(uA).max())      # (weights taken away again)
    seen = _spy(monkeypatch)
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, W=W, algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(ops.prox_unity_plus, axis=0), max_iter=6, e_rel=1e-4, check_convergence=False)
    assert seen == [("f64mfma", "k64_grad_pass")], seen
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=6, e_rel=1e-4, check_convergence=False, W=W)
    _close(A, Ao)
    _close(S, So)
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, A, S, W=W, step=pm.nmf.scaled_step_pgm(0.5), accelerated=True, max_iter=6, e_rel=1e-9)
    Ao, So = A0.copy(), S0.copy()
    orc.pgm_nmf(Y, Ao, So, step=lambda A_, S_, it=None, grads=None: tuple(0.5 * s_ for s_ in orc.lipschitz_steps(A_, S_)), accelerated=True, max_iter=6, e_rel=1e-9, W=W)
    _close(A, Ao)
    _close(S, So)
    with pytest.raises(ValueError):          # the reference's own failure: `if W == 1` on an array (nmf.py:63)
        pm.nmf.nmf(Y, A0.copy(), S0.copy(), W=W, max_iter=2)


@pytest.mark.parametrize("M,N,K,accelerated", [(100, 50, 3, False), (100, 50, 3, True), (700, 900, 40, False), (300, 420, 128, True)])
def test_pgm_with_the_line_search(pm, orc, monkeypatch, M, N, K, accelerated):
    """algorithms.py:110-128 (Beck & Teboulle's backtracking, the reference's own example: examples/unmixing.py:134 on float64 arrays) in fp64:
    factors, the returned gradient, the stopping iteration.  Fixed steps four times the Lipschitz ones force halvings; the first shape is the
    example's (the matrix-core kernels whatever the size: mode "f64mfma")."""
    seen = _spy(monkeypatch)
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, seed=5)
    sA, sS = orc.lipschitz_steps(A0, S0)
    fixed = (4 * sA, 4 * sS)
    f = partial(pm.nmf.log_likelihood, Y=Y)
    A, S = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv, G, steps = pm.nmf.nmf(Y, A, S, step=pm.nmf.constant_step(*fixed), backtracking=True, f=f, accelerated=accelerated, max_iter=12, e_rel=1e-9, callback=tb)
    assert seen[0] == ("f64mfma", "k64_grad_pass"), seen
    Ao, So = A0.copy(), S0.copy()
    trace = []
    oret = orc.pgm_nmf(Y, Ao, So, step=lambda a, s, it, g: fixed, accelerated=accelerated, backtracking=True, max_iter=12, e_rel=1e-9, trace=trace)
    assert len(tb.trace) == len(trace) and tuple(conv) == tuple(oret[0])
    _close(A, Ao, rtol=1e-8)
    _close(S, So, rtol=1e-8)
    _close(G[0], oret[1][0], rtol=1e-7)
    _close(tb.trace[2][0], trace[2][0], rtol=1e-8)
    # chained (no callback): the same bits; the default rule (no halvings, the reference's example) against the oracle
    Ac, Sc = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, Ac, Sc, step=pm.nmf.constant_step(*fixed), backtracking=True, f=f, accelerated=accelerated, max_iter=12, e_rel=1e-9)
    assert np.array_equal(A, Ac) and np.array_equal(S, Sc)
    if not accelerated:
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, backtracking=True, f=f, max_iter=8, e_rel=1e-9)
        Ao, So = A0.copy(), S0.copy()
        orc.pgm_nmf(Y, Ao, So, backtracking=True, max_iter=8, e_rel=1e-9)
        _close(A, Ao, rtol=1e-8)
        _close(S, So, rtol=1e-8)


def test_the_references_example_in_its_own_dtype(pm):
    """examples/unmixing.py: float64 arrays, PGM with backtracking to convergence.  tests/golden/unmixing.npz holds the reference's final loss and
    iteration count; in fp32 the library lands within 0.5 % / 15 % of them (tests/test_gpu_nmf.py), in fp64 on them."""
    from conftest import load_golden
    from test_gpu_nmf import spec_to_prox
    z, meta = load_golden("unmixing.npz")
    Y, A0, S0 = z["Y"].astype(np.float64), z["A0"].astype(np.float64), z["S0"].astype(np.float64)
    done = 0
    for r in meta["runs"]:
        if r["cfg"] is not None or r["mode"] != "nmf":
            continue
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        pm.nmf.nmf(Y, A, S, prox_A=spec_to_prox(pm, tuple(r["prox_A"])), prox_S=spec_to_prox(pm, tuple(r["prox_S"])),
                   backtracking=True, f=partial(pm.nmf.log_likelihood, Y=Y), e_rel=1e-4, max_iter=1000, callback=tb)
        loss = pm.nmf.log_likelihood(A, S, Y=Y)
        if z["Y"].dtype == np.float64:
            assert len(tb.trace) == r["iters"], (len(tb.trace), r["iters"])
            assert loss == pytest.approx(r["loss"], rel=1e-7)
        else:                                  # (a fixture stored in fp32: the fp32 test's bounds)
            assert abs(loss / r["loss"] - 1) < 5e-3 and abs(len(tb.trace) - r["iters"]) <= 0.15 * r["iters"]
        done += 1
    assert done >= 1


@pytest.mark.parametrize("backend", ["pgm", "adaprox", "bsdmm"])
def test_runs_are_bit_reproducible(pm, orc, backend):
    """no atomics, fixed summation order everywhere (slabs folded in order, partial sums folded in order): the same call gives the same bits"""
    ops = pm.operators
    M, N, K = 900, 1300, 72
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=True, seed=12)
    kw = {"pgm": dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)),
          "adaprox": dict(algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(ops.prox_unity_plus, axis=0), check_convergence=False),
          "bsdmm": dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2)}[backend]
    outs = []
    for _ in range(3):
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, max_iter=7, e_rel=1e-6, **kw)
        outs.append((A, S))
    for A, S in outs[1:]:
        assert np.array_equal(A, outs[0][0]) and np.array_equal(S, outs[0][1])


def test_the_helper_functions_keep_float64(pm, orc):
    """nmf.grad_likelihood / log_likelihood / step_pgm called by the caller's own code on float64 arrays (the reference: NumPy in float64)"""
    M, N, K = 700, 900, 48
    Y, A, S = orc.synthetic_problem(M, N, K, np.float64, seed=8)
    W = 0.5 + np.random.default_rng(1).random((M, N))
    for w in (None, W):
        kw = {} if w is None else {"W": w}
        gA, gS = pm.nmf.grad_likelihood(A, S, Y=Y, **kw)
        rA, rS = orc.residual_gradients(A, S, Y, w)
        assert gA.dtype == np.float64
        np.testing.assert_allclose(gA, rA, rtol=1e-12, atol=1e-12 * np.abs(rA).max())
        np.testing.assert_allclose(gS, rS, rtol=1e-12, atol=1e-12 * np.abs(rS).max())
        assert pm.nmf.log_likelihood(A, S, Y=Y, **kw) == pytest.approx(orc.half_sq_residual(A, S, Y, w), rel=1e-12)
    sA, sS = pm.nmf.step_pgm(A, S)
    LA, LS = orc.lipschitz_steps(A, S)
    assert sA == pytest.approx(LA, rel=1e-11) and sS == pytest.approx(LS, rel=1e-11)
    sA2, sS2 = pm.nmf.step_pgm(A, S)                    # the cached factors-only context: a function of its arguments
    assert (sA2, sS2) == (sA, sS)
    s32 = pm.nmf.step_pgm(A.astype(np.float32), S.astype(np.float32))
    assert s32[0] == pytest.approx(LA, rel=1e-5)


def test_a_float64_Y_that_lives_on_the_gpu(pm, orc):
    """a torch float64 tensor (or a pitched view of one) as Y: copied into the fp64 context's array on the device -- the same bits as the host array"""
    import torch
    ops = pm.operators
    M, N, K = 500, 700, 40
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=True, seed=4)
    kw = dict(algorithm=pm.adaprox, scheme="adam", prox_S=partial(ops.prox_unity_plus, axis=0), max_iter=5, e_rel=1e-4, check_convergence=False)
    Ah, Sh = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, Ah, Sh, **kw)
    Yd = torch.from_numpy(Y).to("cuda:0")
    Ad, Sd = A0.copy(), S0.copy()
    pm.nmf.nmf(Yd, Ad, Sd, **kw)
    assert Ad.dtype == np.float64 and np.array_equal(Ah, Ad) and np.array_equal(Sh, Sd)
    wide = torch.zeros((M, N + 24), device="cuda:0", dtype=torch.float64)
    wide[:, :N] = Yd
    Ap, Sp = A0.copy(), S0.copy()
    pm.nmf.nmf(wide[:, :N], Ap, Sp, **kw)
    assert np.array_equal(Ah, Ap) and np.array_equal(Sh, Sp)
    Y2, A2, S2 = orc.synthetic_problem(120, 200, 4, np.float64, seed=4)                   # a small problem: the fused small-problem kernels, same entry
    Ah, Sh = A2.copy(), S2.copy()
    pm.nmf.nmf(Y2, Ah, Sh, max_iter=5)
    Ad, Sd = A2.copy(), S2.copy()
    pm.nmf.nmf(torch.from_numpy(Y2).to("cuda:0"), Ad, Sd, max_iter=5)
    assert np.array_equal(Ah, Ad) and np.array_equal(Sh, Sd)
    with pytest.raises(NotImplementedError):                                               # what the fp64 kernels do not cover has no float64 device path
        pm.nmf.nmf(Yd, A0.copy(), S0.copy(), prox_A=lambda X, step: np.maximum(X, 0), max_iter=2)


def test_switching_the_large_path_off_restores_the_fp32_computation_and_its_warning(pm, orc, monkeypatch, caplog):
    import logging
    from proxmin_amd import algorithms
    from proxmin_amd.engine import DeviceNMF, f64_applies
    monkeypatch.setenv("PMX_F64_BIG", "0")
    assert not f64_applies(256, 512, 64) and f64_applies(200, 1000, 5)
    with pytest.raises(NotImplementedError):
        DeviceNMF(256, 512, 64, mode="f64")
    algorithms._f64_warned.clear()
    Y, A0, S0 = orc.synthetic_problem(256, 512, 64, np.float64, seed=5)
    with caplog.at_level(logging.WARNING, logger="proxmin"):
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, max_iter=2)
    assert [r for r in caplog.records if "float64 arrays" in r.getMessage()]
    assert A.dtype == np.float64
