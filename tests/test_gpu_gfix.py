"""[r5] Mode f16x2r's correction (proxmin_amd/csrc/k_gfix.hip): K1 <HH> forms its residual from the high x high fp16 product alone and what
that leaves out -- A S - a0 s0 = A s_r + a_r s0 -- reaches both gradients through K x K matrices, as one more gradient slab.
Reference: nmf.grad_likelihood (proxmin/nmf.py:28-41).  The arithmetic itself is pinned by tests/test_gpu_kernels.py
(test_mode_f16x2r_removes_the_coherent_error) and the solver runs of test_gpu_parity_*.py; here: the properties of the construction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("K", [64, 128])
def test_exactly_representable_inputs_give_exact_fp32s_bits(eng, K):
    """Small-integer factors and data: every product and every partial sum is exact in every arithmetic (all sums stay below 2^24).
    All modes -- exact fp32, two-term fp16, <R3> (K = 64), <HH> + correction (whose correction is then
    exactly zero in A's high terms ... and exactly what is missing otherwise) -- must return the SAME BITS."""
    M, N = 4096, 2048
    rng = np.random.default_rng(3)
    A = rng.integers(0, 8, (M, K)).astype(np.float32)        # 3 bits: a0 = A, a_r = 0
    S = rng.integers(0, 4, (K, N)).astype(np.float32)
    Y = (A.astype(np.float64) @ S.astype(np.float64) - rng.integers(1, 800, (M, N))).astype(np.float32)
    out = {}
    for mode in ("f32", "f16x2", "f16x2r"):
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y)
            dev.set_factors(A, S)
            out[mode] = dev.grad()
            if mode == "f16x2r":
                assert dev.k1_info()["kernel"] in ("k_grad_f16_v8_hh", "k_grad_f16_k128_hh")
    R = A.astype(np.float64) @ S.astype(np.float64) - Y
    exact = (R @ S.astype(np.float64).T, A.astype(np.float64).T @ R)
    assert max(np.abs(exact[0]).max(), np.abs(exact[1]).max()) < 2 ** 24
    for mode in ("f32", "f16x2", "f16x2r"):
        assert np.array_equal(out[mode][0], exact[0]) and np.array_equal(out[mode][1], exact[1]), mode
    # ... and with factors that DO have low terms (13 significant bits) the correction is what makes <HH> agree with fp64
    A2 = (A + rng.integers(0, 256, (M, K)) / 256.0).astype(np.float32)
    S2 = (S + rng.integers(0, 256, (K, N)) / 1024.0).astype(np.float32)
    r64 = ((A2.astype(np.float64) @ S2.astype(np.float64) - Y) @ S2.astype(np.float64).T, A2.astype(np.float64).T @ (A2.astype(np.float64) @ S2.astype(np.float64) - Y))
    with eng.DeviceNMF(M, N, K, mode="f16x2r") as dev:
        dev.set_Y(Y)
        dev.set_factors(A2, S2)
        gA, gS = dev.grad()
    np.testing.assert_allclose(gA, r64[0], rtol=3e-6, atol=3e-6 * np.abs(r64[0]).max())
    np.testing.assert_allclose(gS, r64[1], rtol=3e-6, atol=3e-6 * np.abs(r64[1]).max())


@pytest.mark.parametrize("M,N,K", [(1024, 1536, 64), (1152, 1024, 128), (1000, 1500, 50)])
def test_one_block_passes_and_repeatability(eng, orc, M, N, K):
    """bsdmm asks K1 for ONE block's gradient at a time (nmf.py:181-185 evaluates both and keeps one): the correction of the block that is
    wanted is computed, the other block's slab is left alone; results equal the two-block pass bit for bit, and two passes agree bit for bit."""
    import ctypes as C
    from proxmin_amd import _lib
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=11)
    r64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    with eng.DeviceNMF(M, N, K, mode="f16x2r") as dev:
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
        gA2, gS2 = dev.grad()
        assert np.array_equal(gA, gA2) and np.array_equal(gS, gS2)
        ms = C.c_double()
        _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 0, 1, C.byref(ms)))      # gA only (twice: warm-up + 1)
        _lib.check(dev.lib.pmx_time_grad(dev.h, 0, 1, 1, C.byref(ms)))      # gS only
        gA3, gS3 = dev.grad()
        assert np.array_equal(gA, gA3) and np.array_equal(gS, gS3)
    for g, r in ((gA, r64[0]), (gS, r64[1])):
        err = float(np.sqrt(((g - r) ** 2).mean()) / np.abs(r).max())
        assert err < 4e-7, err
