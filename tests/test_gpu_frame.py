"""GPU: ragged shapes and in-between K on K1's zero-padded FRAME (include/pmx.h: pmx_k1_frame; pmx_api.hip: choose_frame).

The producer / consumer K1s take M % 128 = 0 and N % 256 = 0 (N % 128 at K = 128).  A ragged M x N problem whose K has such a kernel
runs it on M and N rounded up, with Y (and W) in a zero-padded copy, factor arrays whose extra rows are zero and gradient slabs whose
extra rows nobody reads.  What must hold: the results are those of the M x N problem -- gradients, loss and whole solver runs against
the oracle on the REAL shape, at the tolerances the aligned shapes are held to (reference: proxmin/nmf.py:13-65 takes any shape)."""
import os
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5                      # the north star's tolerance

# (M, N, K, mode) -> the frame and the kernel it buys
FRAMED = [
    (1000, 1500, 64, "f16x2", (1024, 1536), "k_grad_f16_v8"),
    (1000, 1500, 64, "bf16x3", (1024, 1536), "k_grad_bf16"),
    (1000, 1500, 64, "f32", (1024, 1536), "k_grad_f32_pc"),
    (1100, 2000, 128, "f16x2", (1152, 2048), "k_grad_f16_k128"),
    (1000, 1400, 32, "f16x2", (1024, 1536), "k_grad_f16_k32"),
    (1000, 1400, 32, "f32", (1024, 1536), "k_grad_f32_pc"),
    (4000, 5000, 64, "f16x2", (4096, 5120), "k_grad_f16_v8"),
    (16383, 4097, 64, "f16x2", (16384, 4352), "k_grad_f16_v8"),
]
# shapes that must NOT be framed: padding too expensive, no tuned kernel for that K / mode, already aligned, small problem
UNFRAMED = [(300, 260, 64, "f16x2"), (1100, 2000, 128, "bf16x3"), (1000, 1400, 32, "bf16x3"), (1000, 1400, 20, "bf16x3"),
            (1024, 1536, 64, "f16x2"), (200, 1000, 5, "f32"), (1100, 2000, 128, "f32"), (1100, 2000, 100, "f32"), (300, 260, 50, "f16x2")]


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as g
    g.build()
    from proxmin_amd import engine
    return engine


@pytest.fixture(scope="module")
def pm():
    import __graft_entry__ as g
    g.build()
    import proxmin_amd
    proxmin_amd.set_default_mode("f32")
    yield proxmin_amd
    proxmin_amd.set_default_mode(None)


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


def _check_grad(orc, dev, A, S, Y, W=None):
    gA, gS = dev.grad()
    loss = dev.loglike()
    x64 = [x.astype(np.float64) for x in (A, S, Y)] + ([W.astype(np.float64)] if W is not None else [])
    rA, rS = orc.residual_gradients(*x64)
    assert gA.shape == A.shape and gS.shape == S.shape
    np.testing.assert_allclose(gA, rA, rtol=2e-5, atol=2e-5 * np.abs(rA).max())
    np.testing.assert_allclose(gS, rS, rtol=2e-5, atol=2e-5 * np.abs(rS).max())
    assert loss == pytest.approx(orc.half_sq_residual(*x64), rel=2e-5)
    return gA, gS


@pytest.mark.parametrize("M,N,K,mode,frame,kernel", FRAMED)
def test_framed_gradient_matches_oracle(eng, orc, M, N, K, mode, frame, kernel):
    """nmf.grad_likelihood / log_likelihood (nmf.py:13-41) of a ragged problem through the tuned kernel on its frame; repeated
    launches bit-identical; with weights (zeros included) where the kernel takes them."""
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=M + N + K)
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        info = dev.k1_info()
        # `frame` is M and N rounded up to the kernel's tile; the context may take a slightly larger one (<= 6 % more entries) when
        # that makes the chained accumulation of gA possible (pmx_api.hip: pmx_ctx_create)
        got = info["frame"]
        assert info["kernel"] == kernel and got[0] >= frame[0] and got[1] >= frame[1] and got[0] * got[1] <= 1.06 * frame[0] * frame[1], info
        assert got == frame or info["chain"] > 0, info
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = _check_grad(orc, dev, A, S, Y)
        gA2, gS2 = dev.grad()
        assert np.array_equal(gA, gA2) and np.array_equal(gS, gS2)
        assert dev.k1_info()["kernel"] == kernel              # (what ran, not what was planned)
        if kernel != "k_grad_f16_k32":                        # (k_grad_f16_k32 takes no weights)
            rng = np.random.default_rng(8)
            W = (0.1 + 2.0 * rng.random((M, N))).astype(np.float32)
            W[rng.random((M, N)) < 0.15] = 0
            dev.set_W(W)
            _check_grad(orc, dev, A, S, Y, W)
            dev.set_W(None)
        # other factors in the same context: the rows behind M / N stay zero whatever is uploaded
        A2 = (A * 3.0 + 0.25).astype(np.float32)
        S2 = (S * 0.5 + 0.125).astype(np.float32)
        dev.set_factors(A2, S2)
        _check_grad(orc, dev, A2, S2, Y)


@pytest.mark.parametrize("M,N,K,mode", UNFRAMED)
def test_shapes_that_keep_their_own_frame(eng, M, N, K, mode):
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["frame"] == (M, N) and dev.k1_info()["frame_K"] == K, dev.k1_info()


def test_frame_switch_and_device_y(eng, orc):
    """PMX_FRAME=0 keeps the guarded kernels (same results within the K1 tolerance); a DEVICE Y handed over with copy=False is
    copied into the frame (pmx.h: the caller's buffer is not referenced afterwards), with a row pitch larger than N too."""
    import torch
    M, N, K = 1000, 1500, 64
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=3)
    with eng.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = _check_grad(orc, dev, A, S, Y)
    os.environ["PMX_FRAME"] = "0"
    try:
        with eng.DeviceNMF(M, N, K, mode="f16x2") as dev:
            info = dev.k1_info()
            assert info["frame"] == (M, N), info
            dev.set_Y(Y)
            dev.set_factors(A, S)
            hA, hS = _check_grad(orc, dev, A, S, Y)
    finally:
        del os.environ["PMX_FRAME"]
    np.testing.assert_allclose(gA, hA, rtol=4e-5, atol=4e-5 * np.abs(hA).max())
    np.testing.assert_allclose(gS, hS, rtol=4e-5, atol=4e-5 * np.abs(hS).max())
    ld = N + 36
    buf = torch.zeros((M, ld), dtype=torch.float32, device="cuda:0")
    buf[:, :N] = torch.from_numpy(Y).to("cuda:0")
    buf[:, N:] = 7.0                                          # (garbage behind the rows: must not be read)
    with eng.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y_device(buf.data_ptr(), ld=ld, copy=False, keepalive=buf)
        buf.fill_(-1.0)                                       # the context owns a copy
        torch.cuda.synchronize()
        dev.set_factors(A, S)
        dA, dS = _check_grad(orc, dev, A, S, Y)
    assert np.array_equal(dA, gA) and np.array_equal(dS, gS)


CASES = [
    ("pgm", dict()),
    ("fista", dict(accelerated=True)),
    ("adam", dict(scheme="adam")),
    ("bsdmm", dict()),
    ("amsgrad_unity", dict(scheme="amsgrad")),
]


@pytest.mark.parametrize("M,N,K,mode", [(1000, 1500, 64, "f16x2"), (1000, 1500, 64, "f32"), (1100, 2000, 128, "f16x2"), (1000, 1400, 32, "f16x2"),
                                      (1000, 1500, 64, "f16x2r"), (1100, 2000, 128, "f16x2r")])      # [r5] <HH> + the correction slab on a frame
@pytest.mark.parametrize("name,kw", CASES)
def test_framed_solvers_at_rtol_1e4(pm, orc, name, kw, M, N, K, mode):
    """Six iterations of every back-end on a framed problem against the fp64 oracle on the real shape, from identical fp32 inputs:
    smooth back-ends every entry within |x - x_ref| <= 1e-5 + 1e-4 |x_ref| (the bound the aligned shapes are held to in
    test_gpu_parity_strict.py); amsgrad + prox_unity_plus (eps clamp) the same floor as there."""
    from test_gpu_parity_strict import SMOOTH, _solve_pair, frac_within
    unity = name.endswith("unity")
    if unity and K == 32 and mode == "f16x2":
        pytest.skip("covered at K = 64 / 128")
    from proxmin_amd.engine import DeviceNMF
    with DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["frame"] != (M, N)
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=21)
    pm.set_default_mode(mode)
    try:
        A, S, Ao, So = _solve_pair(pm, orc, name, kw, Y, A0, S0, unity, np.float64)
    finally:
        pm.set_default_mode(None)
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    if name.startswith(SMOOTH):
        assert fA == 1.0 and fS == 1.0, "%s %s: %.6f / %.6f within rtol 1e-4 (worst %.1f x)" % (mode, name, fA, fS, max(wA, wS))
    else:
        assert fA >= 0.9995 and fS >= 0.998, (fA, fS, wA, wS)


# ---- K between the tuned ones: K1 runs the next tuned K on zero-padded copies of the factors --------------------------------------
KFRAMED = [
    # (M, N, K, mode) -> K1's K, kernel
    (1024, 1536, 50, "f16x2", 64, "k_grad_f16_v8"),
    (1024, 1536, 50, "bf16x3", 64, "k_grad_bf16"),
    (1024, 1536, 50, "f32", 64, "k_grad_f32_pc"),
    (1000, 1500, 40, "f16x2", 64, "k_grad_f16_v8"),        # ragged rows / columns AND components
    (2048, 1024, 20, "f16x2", 32, "k_grad_f16_k32"),
    (2048, 1024, 20, "f32", 32, "k_grad_f32_pc"),
    (1100, 2000, 100, "f16x2", 128, "k_grad_f16_k128"),
    (2048, 2048, 9, "f16x2", 32, "k_grad_f16_k32"),         # (more than 2^20 entries: not the small-problem path)
]


@pytest.mark.parametrize("M,N,K,mode,Kk,kernel", KFRAMED)
def test_k_framed_gradient_matches_oracle(eng, orc, M, N, K, mode, Kk, kernel):
    """nmf.grad_likelihood / log_likelihood with a K that has no tuned kernel, through the next tuned K's kernel: the oracle on the
    real K; weights; repeated launches bit-identical; PMX_FRAME=0 (the guarded kernels) agrees within the K1 tolerance"""
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=M + N + K)
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        info = dev.k1_info()
        assert info["frame_K"] == Kk and info["kernel"] == kernel, info
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = _check_grad(orc, dev, A, S, Y)
        gA2, gS2 = dev.grad()
        assert np.array_equal(gA, gA2) and np.array_equal(gS, gS2)
        if kernel != "k_grad_f16_k32":
            rng = np.random.default_rng(8)
            W = (0.1 + 2.0 * rng.random((M, N))).astype(np.float32)
            W[rng.random((M, N)) < 0.15] = 0
            dev.set_W(W)
            _check_grad(orc, dev, A, S, Y, W)
            dev.set_W(None)
        A2 = (A * 3.0 + 0.25).astype(np.float32)
        S2 = (S * 0.5 + 0.125).astype(np.float32)
        dev.set_factors(A2, S2)
        _check_grad(orc, dev, A2, S2, Y)
    os.environ["PMX_FRAME"] = "0"
    try:
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            assert dev.k1_info()["frame_K"] == K
            dev.set_Y(Y)
            dev.set_factors(A, S)
            hA, hS = _check_grad(orc, dev, A, S, Y)
    finally:
        del os.environ["PMX_FRAME"]
    np.testing.assert_allclose(gA, hA, rtol=4e-5, atol=4e-5 * np.abs(hA).max())
    np.testing.assert_allclose(gS, hS, rtol=4e-5, atol=4e-5 * np.abs(hS).max())


@pytest.mark.parametrize("M,N,K,mode", [(1024, 1536, 50, "f16x2"), (1000, 1500, 40, "f32"), (1100, 2000, 100, "f16x2"), (2048, 1024, 20, "f16x2"),
                                      (1000, 1500, 50, "f16x2r"), (1100, 2000, 100, "f16x2r")])      # [r5] rows, columns AND components padded under <HH>
@pytest.mark.parametrize("name,kw", CASES)
def test_k_framed_solvers_at_rtol_1e4(pm, orc, name, kw, M, N, K, mode):
    """six iterations of every back-end with an in-between K against the fp64 oracle: smooth back-ends every entry within the north
    star's bound, amsgrad + prox_unity_plus the floor of the aligned shapes"""
    from test_gpu_parity_strict import SMOOTH, _solve_pair, frac_within
    unity = name.endswith("unity")
    if unity and K == 20:
        pytest.skip("covered at K = 50 / 100")
    from proxmin_amd.engine import DeviceNMF
    with DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["frame_K"] != K
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=21)
    pm.set_default_mode(mode)
    try:
        A, S, Ao, So = _solve_pair(pm, orc, name, kw, Y, A0, S0, unity, np.float64)
    finally:
        pm.set_default_mode(None)
    fA, wA = frac_within(A, Ao)
    fS, wS = frac_within(S, So)
    if name.startswith(SMOOTH):
        assert fA == 1.0 and fS == 1.0, "%s %s: %.6f / %.6f within rtol 1e-4 (worst %.1f x)" % (mode, name, fA, fS, max(wA, wS))
    else:
        assert fA >= 0.9995 and fS >= 0.998, (fA, fS, wA, wS)


@pytest.mark.parametrize("accel", [False, True])
def test_framed_line_search(pm, orc, accel):
    """algorithms.py:110-127 on a framed problem: the trial points (Xe / X_ buffers) are K1 inputs as well.  A 1.5 x too long
    fixed step forces halvings; against the fp64 oracle and against the same run on the guarded kernels (PMX_FRAME=0)."""
    M, N, K = 600, 700, 64                                    # frame 640 x 768 (17 % more entries); small enough for the host oracle
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=17)
    from proxmin_amd.engine import DeviceNMF
    with DeviceNMF(M, N, K, mode="f32") as dev:
        assert dev.k1_info()["frame"] == (640, 768)
    sA, sS = orc.lipschitz_steps(A0.astype(np.float64), S0.astype(np.float64))
    fixed = (1.5 * sA, 1.5 * sS)

    def run():
        A, S = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        pm.nmf.nmf(Y, A, S, step=pm.nmf.constant_step(*fixed), accelerated=accel, backtracking=True,
                   f=partial(pm.nmf.log_likelihood, Y=Y), max_iter=4, e_rel=1e-9, callback=tb)
        return A, S, len(tb.trace)

    A, S, n = run()
    os.environ["PMX_FRAME"] = "0"
    try:
        Au, Su, nu = run()
    finally:
        del os.environ["PMX_FRAME"]
    Ao, So = A0.astype(np.float64), S0.astype(np.float64)
    trace = []
    orc.pgm_nmf(Y.astype(np.float64), Ao, So, step=lambda a, s, it, g: fixed, accelerated=accel, backtracking=True, max_iter=4, e_rel=1e-9, trace=trace)
    assert np.isfinite(Ao).all() and n == len(trace) == nu
    np.testing.assert_allclose(A, Ao, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(S, So, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(A, Au, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(S, Su, rtol=RTOL, atol=ATOL)
