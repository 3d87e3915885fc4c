"""GPU: the range guard of the two-term fp16 kernels (k_grad_f16_v8.hip: f16_range_fault; pmx_api.hip: k1_leave_f16).

Mode f16x2 scales the residual R = A S - Y for fp16 with ONE power of two taken from a bound, max|Y| + K max|A| max|S|.  Where the
model term is far above the data (factors that ran away: RAdam's unrectified first steps with nmf.step_adaprox, algorithms.py:196-213;
a start far from the data) the entries with P = 0 -- rows the prox has set to zero, whose whole gradient is -Y S^T -- fall below
fp16's range under that scale.  The reference computes them in the array's own precision (nmf.py:28-41): fp32 carries an exponent per
entry.  A fuzz run found it (scratch/fuzz_nmf.py, RAdam at K -> 128: 45 % of the entries zero where the oracle's are 1e14).  The
kernels now refuse such a launch before anything is written and the context continues THE SAME iteration with the exact-fp32 K1 of
its frame.  Tested here:
* a first launch that faults: everything that follows is the fp32 kernel's -- gradients, loss and every back-end BIT FOR BIT equal to
  a context created in mode f32 (same frame, same plan), at the shapes of all three fp16 kernels, chained and framed ones included;
* a fault in the middle of a run (RAdam, third iteration): the fp64 oracle at the floor of mode f32;
* weights survive the switch; ordinary problems never trip the guard; PMX_F16_RANGE=0 switches it off."""
import os
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (M, N, K, the fp16 kernel of the shape, the fp32 kernel it falls back to)
SHAPES = [
    (1024, 1024, 64, "k_grad_f16_v8", "k_grad_f32_pc"),
    (4096, 4096, 64, "k_grad_f16_v8", "k_grad_f32_pc"),          # chained (gA summed in place) before and after
    (1408, 1024, 128, "k_grad_f16_k128", "k_grad_f32"),
    (1024, 1024, 32, "k_grad_f16_k32", "k_grad_f32_pc"),
    (1000, 1500, 50, "k_grad_f16_v8", "k_grad_f32_pc"),          # framed in rows, columns and components
    (1100, 2000, 100, "k_grad_f16_k128", "k_grad_f32"),
]


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as g
    g.build()
    from proxmin_amd import engine
    return engine


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


def _far_start(orc, M, N, K, unity=False):
    """a start far above the data: K max|A| max|S| ~ 1e9 max|Y|"""
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=M + K)
    return Y, (A0 * 3e3).astype(np.float32), (S0 * 1e4).astype(np.float32)


def _begin(dev, backend):
    from proxmin_amd import operators as ops
    plus = [ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(ops.prox_plus, 1)]
    if backend == "adaprox":
        dev.adaprox_begin(plus, scheme="amsgrad", e_rel=(1e-3, 1e-3), check_convergence=False)
        return lambda n: dev.adaprox_run(np.full(n, 0.9), 0.9)
    if backend in ("pgm", "fista"):
        dev.pgm_begin(plus, accelerated=backend == "fista", e_rel=(1e-9, 1e-9))
        return dev.pgm_run
    soft = partial(ops.prox_soft, thresh=1e-3)
    dev.bsdmm_begin(plus, [[ops.device_proxseq(ops.prox_plus, j), ops.device_proxseq(soft, j)] for j in range(2)], e_rel=(1e-9, 1e-9))
    return dev.bsdmm_run


def _name(k16, mode):
    """the kernel a mode-f16x2r context runs where a mode-f16x2 context runs k16 ([r5]: <HH> + the correction slab at K1's K = 64 / 128, <R3> at 32)"""
    if mode == "f16x2":
        return k16
    return {"k_grad_f16_v8": "k_grad_f16_v8_hh", "k_grad_f16_k128": "k_grad_f16_k128_hh", "k_grad_f16_k32": "k_grad_f16_k32_r3"}[k16]


@pytest.mark.parametrize("mode16", ["f16x2", "f16x2r"])
@pytest.mark.parametrize("M,N,K,k16,k32", SHAPES)
def test_faulting_gradient_is_the_fp32_kernels(eng, orc, M, N, K, k16, k32, mode16):
    """pmx_grad / pmx_loglike: the first launch faults, the retry inside the same call runs the fp32 kernel of the frame; gradients
    bit for bit those of a mode-f32 context and within the K1 tolerance of the fp64 oracle; the loss-only pass never faults"""
    Y, A, S = _far_start(orc, M, N, K)
    with eng.DeviceNMF(M, N, K, mode=mode16) as dev:
        i0 = dev.k1_info()
        assert i0["kernel"] == _name(k16, mode16) and i0["range_faults"] == 0, i0
        dev.set_Y(Y)
        dev.set_factors(A, S)
        loss16 = dev.loglike()                       # (loss-only instance: fp32 residuals, no guard)
        assert dev.k1_info()["range_faults"] == 0
        gA, gS = dev.grad()
        i1 = dev.k1_info()
        assert i1["kernel"] == k32 and i1["range_faults"] == 1 and i1["frame"] == i0["frame"] and i1["frame_K"] == i0["frame_K"], (i0, i1)
        gA2, gS2 = dev.grad()
        loss = dev.loglike()
        assert dev.k1_info()["range_faults"] == 1
    with eng.DeviceNMF(M, N, K, mode="f32") as dev:
        assert dev.k1_info()["kernel"] == k32
        same_plan = dev.k1_info()["frame"] == i0["frame"] and dev.k1_info()["frame_K"] == i0["frame_K"]   # (mode f32 pads K only up to 64)
        dev.set_Y(Y)
        dev.set_factors(A, S)
        hA, hS = dev.grad()
        hloss = dev.loglike()
    assert np.array_equal(gA, gA2) and np.array_equal(gS, gS2)
    if same_plan:
        assert np.array_equal(gA, hA) and np.array_equal(gS, hS) and loss == hloss
    assert loss == pytest.approx(hloss, rel=2e-5) and loss16 == pytest.approx(hloss, rel=2e-5)
    rA, rS = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    np.testing.assert_allclose(gA, rA, rtol=2e-5, atol=2e-5 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=2e-5, atol=2e-5 * np.abs(rS).max())


@pytest.mark.parametrize("mode16", ["f16x2", "f16x2r"])
@pytest.mark.parametrize("backend", ["adaprox", "pgm", "fista", "bsdmm"])
@pytest.mark.parametrize("M,N,K,k16,k32", SHAPES[1:5])
def test_faulting_first_iteration_equals_mode_f32(eng, orc, backend, M, N, K, k16, k32, mode16):
    """every back-end from a start that trips the guard in its first K1 launch: the iteration is repeated with the fp32 kernel and the
    run is the run of a mode-f32 context (iteration count; factors bit for bit under adaprox)"""
    Y, A, S = _far_start(orc, M, N, K)
    out = {}
    for mode in (mode16, "f32"):
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y)
            dev.set_factors(A, S)
            run = _begin(dev, backend)
            r = run(4)
            info = dev.k1_info()
            assert r.iterations == 4 and info["kernel"] == k32 and info["range_faults"] == (1 if mode == mode16 else 0), (mode, r.iterations, info)
            out[mode] = dev.get_factors()
    for a, b in zip(out[mode16], out["f32"]):
        if backend == "adaprox":
            assert np.array_equal(a, b)
        else:      # (the step rule in front of the refused K1 ran twice: its power iteration restarts from the first attempt's vector)
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * np.abs(b).max())


@pytest.mark.parametrize("M,N,K", [(1408, 1024, 128), (1024, 1024, 64), (1024, 1024, 32), (1433, 704, 119)])
def test_radam_runaway_meets_the_oracle(orc, M, N, K):
    """What the fuzz run found.  RAdam's first steps are plain momentum steps of size alpha (algorithms.py:196-213): with
    nmf.step_adaprox the factors of a unit-scale problem reach 1e6 in two iterations and 1e14 in three.  The third K1 launch sees
    K max|A| max|S| ~ 2^40 max|Y|: mode f16x2 continues in fp32 from there and meets the fp64 oracle like mode f32 does (before the
    guard: 40-45 % of the entries zero where the oracle's are huge)."""
    import proxmin_amd as pm
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=5)
    Ao, So = A0.astype(np.float64), S0.astype(np.float64)
    orc.adaprox_nmf(Y.astype(np.float64), Ao, So, ("plus",), ("plus",), scheme="radam", max_iter=3, e_rel=1e-3, check_convergence=False)
    assert np.abs(Ao).max() > 1e10
    frac = {}
    try:
        for mode in ("f32", "f16x2"):
            pm.set_default_mode(mode)
            A, S = A0.copy(), S0.copy()
            pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="radam", max_iter=3, e_rel=1e-3, check_convergence=False)
            f = []
            for a, b in ((A, Ao), (S, So)):
                assert np.isfinite(a).all()
                f.append(float((np.abs(a - b) <= 2e-5 + 2e-4 * np.abs(b)).mean()))
            frac[mode] = min(f)
    finally:
        pm.set_default_mode(None)
    assert frac["f32"] >= 0.999 and frac["f16x2"] >= 0.999, frac


def test_weights_survive_the_switch(eng, orc):
    M, N, K = 1408, 1024, 128
    Y, A, S = _far_start(orc, M, N, K)
    rng = np.random.default_rng(3)
    W = (0.1 + 2.0 * rng.random((M, N))).astype(np.float32)
    W[rng.random((M, N)) < 0.2] = 0
    with eng.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y(Y)
        dev.set_W(W)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
        assert dev.k1_info()["range_faults"] == 1 and dev.k1_info()["kernel"] == "k_grad_f32"
    rA, rS = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64), W.astype(np.float64))
    np.testing.assert_allclose(gA, rA, rtol=2e-5, atol=2e-5 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=2e-5, atol=2e-5 * np.abs(rS).max())


def test_ordinary_problems_never_trip_the_guard(eng, orc, monkeypatch):
    """the bench's kind of problem through 12 AMSGrad iterations, and factors scaled apart by 1e3 / 1e-3 (same product): no fault;
    a start 300 x above the data (ratio ~ 2^15): still fp16; PMX_F16_RANGE=0: no guard at any ratio"""
    from proxmin_amd import operators as ops
    M, N, K = 1024, 1536, 64
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=11)
    with eng.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y(Y)
        dev.set_factors(A0, S0)
        dev.adaprox_begin([ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_unity_plus, axis=0), 1)], scheme="amsgrad", e_rel=(1e-3, 1e-3))
        dev.adaprox_run(np.full(12, 0.9), 0.9)
        assert dev.k1_info()["range_faults"] == 0 and dev.k1_info()["kernel"] == "k_grad_f16_v8"
        for fa, fs in ((1e3, 1e-3), (1e-3, 1e3), (20.0, 15.0)):
            dev.set_factors((A0 * fa).astype(np.float32), (S0 * fs).astype(np.float32))
            dev.grad()
            assert dev.k1_info()["range_faults"] == 0, (fa, fs)
    Y, A, S = _far_start(orc, M, N, K)
    monkeypatch.setenv("PMX_F16_RANGE", "0")
    with eng.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y(Y)
        dev.set_factors(A, S)
        dev.grad()
        assert dev.k1_info()["range_faults"] == 0 and dev.k1_info()["kernel"] == "k_grad_f16_v8"


@pytest.mark.parametrize("case", ["user_prox_pgm", "user_step_pgm", "line_search", "user_prox_adaprox", "user_prox_bsdmm"])
def test_one_iteration_per_call_paths_switch_on_the_spot(orc, case):
    """Paths that run one iteration (or a piece of one) per call -- user callables (algorithms.py:37-39, 73-77), the line search
    (:110-127) -- have no iteration to repeat: there every fp16 K1 launch is awaited and, refused, repeated in exact fp32 before
    anything else is enqueued (pmx_api.hip: enqueue_grad).  From a start that trips the guard at once, mode f16x2 gives what mode f32
    gives, bit for bit (the callables are the library's operators written out in NumPy)."""
    import proxmin_amd as pm
    M, N, K = 1024, 1024, 64
    Y, A0, S0 = _far_start(orc, M, N, K)

    def my_plus(X, step):
        return np.maximum(X, 0)

    def my_step(*X, it=None):
        return tuple(0.5 * s for s in pm.nmf.step_pgm(*X))

    out = {}
    try:
        for mode in ("f16x2", "f32"):
            pm.set_default_mode(mode)
            A, S = A0.copy(), S0.copy()
            if case == "user_prox_pgm":
                pm.nmf.nmf(Y, A, S, prox_A=my_plus, max_iter=3, e_rel=1e-12)
            elif case == "user_step_pgm":
                pm.nmf.nmf(Y, A, S, step=my_step, max_iter=3, e_rel=1e-12)
            elif case == "line_search":
                pm.nmf.nmf(Y, A, S, backtracking=True, f=partial(pm.nmf.log_likelihood, Y=Y), max_iter=3, e_rel=1e-12)
            elif case == "user_prox_adaprox":
                pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="amsgrad", prox_S=my_plus, max_iter=3, e_rel=1e-3, check_convergence=False)
            else:
                pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, prox_A=my_plus, proxs_g=[[pm.operators.prox_plus], None], max_iter=3, e_rel=1e-12)
            out[mode] = (A, S)
    finally:
        pm.set_default_mode(None)
    for a, b in zip(out["f16x2"], out["f32"]):
        assert np.isfinite(a).all()
        if case in ("user_prox_adaprox",):
            assert np.array_equal(a, b)
        else:      # (pgm / bsdmm: the step rule in front of the refused K1 ran twice, see above)
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * np.abs(b).max())
