"""GPU: round 6's host-side contract changes.

* the LIBRARY'S default arithmetic is the benchmarked one (mode f16x2r): a user who swaps the import gets the headline's kernels;
* nmf() adopts a Y that already lives in HBM (anything with __cuda_array_interface__) in place: same bits as the host-array call;
* fp64 arrays that the fp64 kernels do not take are computed in fp32 -- and the library SAYS so, once (the reference keeps fp64, nmf.py:39-41);
* the stand-alone Barzilai-Borwein sums propagate NaN like np.max does (utils.py:222);
* bench.py's measurement skeleton (libpmx_floor.so) loads and measures.
"""
import ctypes as C
import logging
import os
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pm():
    import __graft_entry__ as g
    g.build()
    import proxmin_amd
    proxmin_amd.set_default_mode(None)
    return proxmin_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import nmf_oracle
    return nmf_oracle


def test_the_library_default_is_the_benchmarked_arithmetic(pm):
    from proxmin_amd.engine import DeviceNMF
    assert pm.LIBRARY_DEFAULT_MODE == "f16x2r"
    if not os.environ.get("PMX_MODE"):
        assert pm.get_default_mode() == "f16x2r"
    for (M, N, K), kernel in (((256, 512, 64), "k_grad_f16_v8_hh"), ((256, 256, 128), "k_grad_f16_k128_hh"), ((256, 256, 32), "k_grad_f16_k32_r3")):
        with DeviceNMF(M, N, K) as dev:          # no mode argument, no set_default_mode(): what nmf() opens
            assert dev.mode == pm.get_default_mode()
            if dev.mode == "f16x2r":
                assert dev.k1_info()["kernel"] == kernel, dev.k1_info()


@pytest.mark.parametrize("backend", ["pgm", "adaprox", "bsdmm"])
def test_nmf_adopts_a_device_resident_Y_in_place(pm, orc, backend):
    """Y as a torch tensor on the GPU (never copied to the host) against the same call with the host array: the same kernels on the same
    bytes -- bit-identical factors; and against the oracle at the module's usual tolerance."""
    import torch
    M, N, K = 384, 512, 64
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=(backend == "adaprox"), seed=11)
    Yd = torch.from_numpy(Y).to("cuda:0")
    kw = {"pgm": {}, "bsdmm": dict(algorithm=pm.bsdmm),
          "adaprox": dict(algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(pm.operators.prox_unity_plus, axis=0))}[backend]
    Ah, Sh = A0.copy(), S0.copy()
    pm.nmf.nmf(Y, Ah, Sh, max_iter=6, e_rel=1e-9, **kw)
    Ad, Sd = A0.copy(), S0.copy()
    pm.nmf.nmf(Yd, Ad, Sd, max_iter=6, e_rel=1e-9, **kw)
    assert np.array_equal(Ah, Ad) and np.array_equal(Sh, Sd)
    # the helper functions take it as well
    g_h = pm.nmf.grad_likelihood(A0, S0, Y=Y)
    g_d = pm.nmf.grad_likelihood(A0, S0, Y=Yd)
    assert np.array_equal(g_h[0], g_d[0]) and np.array_equal(g_h[1], g_d[1])
    assert pm.nmf.log_likelihood(A0, S0, Y=Yd) == pm.nmf.log_likelihood(A0, S0, Y=Y)
    # a pitched view (rows of a wider tensor): adopted with its row pitch
    wide = torch.zeros((M, N + 64), device="cuda:0", dtype=torch.float32)
    wide[:, :N] = Yd
    Ap, Sp = A0.copy(), S0.copy()
    pm.nmf.nmf(wide[:, :N], Ap, Sp, max_iter=6, e_rel=1e-9, **kw)
    assert np.array_equal(Ah, Ap) and np.array_equal(Sh, Sp)
    with pytest.raises(NotImplementedError):
        pm.nmf.nmf(Yd.double(), A0.copy(), S0.copy(), max_iter=1)          # a float64 device Y goes with float64 factors (the fp64 kernels: tests/test_gpu_f64_big.py)


def test_float64_arrays_computed_in_float32_are_announced_once(pm, orc, caplog, monkeypatch):
    from proxmin_amd import algorithms
    monkeypatch.setenv("PMX_F64_BIG", "0")          # (with the large fp64 kernels on, this problem is computed in fp64: tests/test_gpu_f64_big.py)
    algorithms._f64_warned.clear()
    Y, A0, S0 = orc.synthetic_problem(256, 512, 64, np.float64, seed=5)     # K = 64: outside the small fp64 kernels (K <= 16)
    with caplog.at_level(logging.WARNING, logger="proxmin"):
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, max_iter=2)
        pm.nmf.nmf(Y, A0.copy(), S0.copy(), max_iter=2)
    msgs = [r.getMessage() for r in caplog.records if "float64 arrays" in r.getMessage()]
    assert len(msgs) == 1 and "float32" in msgs[0], msgs
    assert A.dtype == np.float64          # cast back into the caller's arrays
    # a small fp64 problem runs the fp64 kernels: no message
    algorithms._f64_warned.clear()
    caplog.clear()
    Y, A0, S0 = orc.synthetic_problem(64, 96, 4, np.float64, seed=5)
    with caplog.at_level(logging.WARNING, logger="proxmin"):
        pm.nmf.nmf(Y, A0.copy(), S0.copy(), max_iter=2)
    assert not [r for r in caplog.records if "float64 arrays" in r.getMessage()]


def test_barzilai_borwein_sums_propagate_nan(pm):
    """utils.py:222 `r * np.max(np.abs(X)) / np.max(np.abs(G))`: a NaN in X or G gives a NaN step in the reference (np.max propagates it);
    fmax() on the device used to drop it and hand a diverged run a finite step."""
    rng = np.random.default_rng(3)
    X = [rng.random((50, 7)), rng.random((7, 60))]
    G = [rng.standard_normal((50, 7)), rng.standard_normal((7, 60))]
    G[0][17, 3] = np.nan
    st = pm.utils.BarzilaiBorweinStepper(type=1, init_r=0.1)
    s = st.step(*X, it=0, grads=G)
    assert np.isnan(s[0]) and np.isfinite(s[1])
    want = 0.1 * np.max(np.abs(X[1])) / np.max(np.abs(G[1]))
    np.testing.assert_allclose(s[1], want, rtol=1e-12)
    # repeated calls reuse the library's scratch buffer (no allocation per call): same numbers
    for _ in range(3):
        s2 = pm.utils.BarzilaiBorweinStepper(type=1, init_r=0.1).step(*X, it=0, grads=G)
        assert np.isnan(s2[0]) and s2[1] == s[1]


def test_the_measurement_skeleton_loads_and_measures(pm):
    lib = C.CDLL(os.path.join(ROOT, "proxmin_amd", "libpmx_floor.so"))
    lib.pmxf_last_error.restype = C.c_char_p
    v = C.c_double()
    assert lib.pmxf_copy(0, C.c_int64(1 << 28), 3, C.byref(v)) == 0, lib.pmxf_last_error()
    assert 1000.0 < v.value < 9000.0, v.value                         # GB/s read + written: an MI355X streams several TB/s
    assert lib.pmxf_stream(0, 1011, 4096, 4096, 0, 3, C.byref(v)) == 0, lib.pmxf_last_error()
    assert 0.0 < v.value < 5.0, v.value
    assert lib.pmxf_stream(0, 77, 4096, 4096, 0, 3, C.byref(v)) != 0  # an unknown variant is refused
    assert lib.pmxf_mfma(0, 1, 2, C.byref(v)) == 0, lib.pmxf_last_error()
    assert 300.0 < v.value < 2600.0, v.value                          # TFLOP/s


@pytest.mark.parametrize("mode", ["f32", "f16x2r"])
@pytest.mark.parametrize("accelerated", [False, True])
def test_the_gram_fold_riding_in_k1_is_bit_identical_to_the_reduce_launch(pm, orc, mode, accelerated):
    """pgm at K = 32 (cfg2's kernels: k_grad_f32_pc<32>, k_grad_f16_k32): the step rule's Gram fold and the previous iteration's stopping test ride in K1's first
    workgroups (pmx_common.h: k1_gram_fold; PMX_FOLD_IN_K1, read at context creation) -- same arithmetic in the same order as k_gram_reduce: identical factors,
    gradients, steps and stopping iteration, also when the stopping test fires inside the run."""
    from proxmin_amd.engine import DeviceNMF
    M, N, K = 2048, 4096, 32
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=21)
    with DeviceNMF(M, N, K, mode=mode) as dev:
        assert dev.k1_info()["kernel"] == ("k_grad_f32_pc" if mode == "f32" else "k_grad_f16_k32_r3")
    kw = dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)) if accelerated else {}
    out = {}
    pm.set_default_mode(mode)
    try:
        for fold in ("1", "0"):
            os.environ["PMX_FOLD_IN_K1"] = fold
            for e_rel, its in ((1e-12, 17), (3e-2, 300)):          # a fixed count / a run the stopping test ends
                A, S = A0.copy(), S0.copy()
                conv, G, steps = pm.nmf.nmf(Y, A, S, max_iter=its, e_rel=e_rel, **kw)            # chunks of iterations enqueued ahead: the ride's home
                Ac, Sc = A0.copy(), S0.copy()
                tb = pm.utils.Traceback()
                pm.nmf.nmf(Y, Ac, Sc, max_iter=its, e_rel=e_rel, callback=tb, **kw)              # one iteration per call (the callback wants every iterate)
                assert np.array_equal(A, Ac) and np.array_equal(S, Sc), "chained and per-iteration runs differ (fold %s, e_rel %g)" % (fold, e_rel)
                out[(fold, e_rel)] = (A, S, G, steps, conv, len(tb.trace))
    finally:
        os.environ.pop("PMX_FOLD_IN_K1", None)
        pm.set_default_mode(None)
    for e_rel in (1e-12, 3e-2):
        a, b = out[("1", e_rel)], out[("0", e_rel)]
        assert a[5] == b[5] and a[4] == b[4], (a[4:], b[4:])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1])
        assert tuple(a[3]) == tuple(b[3])
    assert 1 < out[("1", 3e-2)][5] < 300, "the second run is meant to stop on its test (%d iterations)" % out[("1", 3e-2)][5]
    # and against the oracle (the module's usual bound)
    A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
    step = (lambda A_, S_, it, grads: tuple(0.5 * s_ for s_ in orc.lipschitz_steps(A_, S_))) if accelerated else None
    orc.pgm_nmf(Y.astype(np.float64), A64, S64, max_iter=17, e_rel=1e-12, accelerated=accelerated, step=step)
    np.testing.assert_allclose(out[("1", 1e-12)][0], A64, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out[("1", 1e-12)][1], S64, rtol=1e-4, atol=1e-5)
