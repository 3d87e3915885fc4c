"""CPU: the C-ABI library builds, loads, and exports every symbol include/pmx.h declares."""
import ctypes
import os
import re

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from proxmin_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmx_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(os.path.join(ROOT, "proxmin_amd", "libpmx.so"))
    for n in names:
        assert hasattr(raw, n), "libpmx.so does not export %s" % n


def test_binding_covers_header(lib):
    from proxmin_amd import _lib
    assert set(_declared_symbols()) == set(_lib.EXPORTED_SYMBOLS)
    assert lib.pmx_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    """sizes implied by include/pmx.h (checked against the ctypes mirrors)."""
    from proxmin_amd import _lib
    assert ctypes.sizeof(_lib.Prox) == 24
    assert ctypes.sizeof(_lib.ProxSeq) == 8 + 24 * _lib.MAX_SEQ
    assert ctypes.sizeof(_lib.Result) == 4 * 5 + 4 + 16 + 16   # 5 ints + pad, 2 doubles, 2 int64
    # and every parameter struct against the compiled library's own sizeof
    sizes = (ctypes.c_int * 5)()
    assert _lib.load().pmx_abi_sizes(sizes) == 0
    assert list(sizes) == [ctypes.sizeof(t) for t in (_lib.ProxSeq, _lib.PgmParams, _lib.AdaproxParams, _lib.BsdmmParams, _lib.Result)]


def test_no_gpu_fails_loudly(lib):
    """Without a GPU every compute entry must refuse (no silent CPU path)."""
    import numpy as np
    from proxmin_amd import _lib, operators
    if lib.pmx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.PmxError):
        operators.prox_plus(np.zeros((4, 4)), 1.0)
    import proxmin_amd as pm
    with pytest.raises(_lib.PmxError):
        pm.nmf.nmf(np.ones((8, 8)), np.ones((8, 2)), np.ones((2, 8)), max_iter=1)


def test_prox_recognition():
    from functools import partial
    from proxmin_amd import operators as ops, _lib
    s = ops.device_proxseq(partial(ops.prox_unity_plus, axis=0), 1)
    assert s.n == 1 and s.seq[0].op == _lib.PROX["unity_plus"] and s.seq[0].unit == 0
    s = ops.device_proxseq(partial(ops.prox_unity_plus, axis=1), 0)
    assert s.seq[0].unit == 0
    s = ops.device_proxseq(partial(ops.prox_unity, axis=0), 0)
    assert s.seq[0].unit == 1
    s = ops.device_proxseq(partial(ops.prox_soft, thresh=0.5, type="absolute"), 0)
    assert s.seq[0].op == _lib.PROX["soft"] and s.seq[0].relative == 0 and abs(s.seq[0].thresh - 0.5) < 1e-7
    assert ops.device_proxseq(None, 0).n == 0
    ap = ops.AlternatingProjections([partial(ops.prox_unity, axis=1), ops.prox_plus], repeat=3)
    s = ops.device_proxseq(ap, 0)
    assert s.n == 2 and s.repeat == 3 and s.seq[0].op == _lib.PROX["plus"] and s.seq[1].op == _lib.PROX["unity"]
    with pytest.raises(NotImplementedError):
        ops.device_proxseq(lambda X, step: X, 0)


def test_weighted_step_rule_dispatch_mirrors_step_pgms_own_argument():
    """nmf.step_pgm tests `W == 1` on ITS OWN W (nmf.py:63): `partial(step_pgm, W=<array>)` -- what nmf() builds for a weighted
    problem (nmf.py:152) -- raises ValueError before anything runs; the bare function and `scaled_step_pgm(c)` (= the idiom
    `lambda *X, it=None: tuple(c * s for s in step_pgm(*X))`) carry W = 1 and must get as far as the device (no GPU here: PmxError)."""
    from functools import partial
    import numpy as np
    import proxmin_amd as pm
    from proxmin_amd import _lib
    Y, A, S = np.ones((8, 8), np.float32), np.ones((8, 2), np.float32), np.ones((2, 8), np.float32)
    W = np.full((8, 8), 2.0, np.float32)
    grad = partial(pm.nmf.grad_likelihood, Y=Y, W=W)
    with pytest.raises(ValueError):
        pm.pgm([A.copy(), S.copy()], grad, partial(pm.nmf.step_pgm, W=W), prox=[pm.operators.prox_plus] * 2, max_iter=1)
    with pytest.raises(ValueError):
        pm.nmf.nmf(Y, A.copy(), S.copy(), W=W, max_iter=1)
    for step in (pm.nmf.step_pgm, pm.nmf.scaled_step_pgm(0.5), partial(pm.nmf.step_pgm, W=1)):
        try:
            pm.pgm([A.copy(), S.copy()], grad, step, prox=[pm.operators.prox_plus] * 2, max_iter=1)
        except _lib.PmxError:
            pass                       # (no GPU in this container: the call got past the step-rule dispatch)


def test_default_mode_and_its_reset():
    """[r6] the library's default arithmetic is the benchmarked one; set_default_mode(None) goes back to it"""
    import proxmin_amd as pm
    assert pm.LIBRARY_DEFAULT_MODE == "f16x2r"
    before = pm.get_default_mode()
    try:
        pm.set_default_mode("f32")
        assert pm.get_default_mode() == "f32"
        pm.set_default_mode(None)
        assert pm.get_default_mode() == os.environ.get("PMX_MODE", "f16x2r")
        with pytest.raises(AssertionError):
            pm.set_default_mode("fp8")
    finally:
        pm.set_default_mode(before)


def test_device_array_detection_is_host_only_logic():
    """engine.as_device_array: NumPy arrays and host objects are host data; an object with __cuda_array_interface__ is adopted with its pitch"""
    from proxmin_amd.engine import as_device_array, DeviceArrayRef
    assert as_device_array(np.zeros((3, 4), np.float32)) is None
    assert as_device_array([[1.0, 2.0]]) is None

    class Fake:
        def __init__(self, typestr="<f4", shape=(6, 8), strides=None):
            self.__cuda_array_interface__ = {"typestr": typestr, "shape": shape, "strides": strides, "data": (4096, False), "version": 3}
    r = as_device_array(Fake())
    assert isinstance(r, DeviceArrayRef) and r.shape == (6, 8) and r.ld == 8 and r.ptr == 4096 and r.dtype == np.float32
    assert as_device_array(Fake(strides=(48, 4))).ld == 12          # rows of a wider array
    r8 = as_device_array(Fake(typestr="<f8", strides=(96, 8)))      # [r6] float64: taken by the fp64 kernels (copied into the context's own array)
    assert r8.dtype == np.float64 and r8.ld == 12
    for bad in (Fake(typestr="<f2"), Fake(typestr="<i4"), Fake(shape=(6,)), Fake(strides=(32, 8)), Fake(strides=(16, 4)), Fake(typestr="<f8", strides=(48, 4))):
        with pytest.raises(TypeError):
            as_device_array(bad)
    with pytest.raises(TypeError):
        np.asarray(r)                                                # never silently copied to the host


def test_sharded_dispatch_refuses_what_the_protocol_does_not_carry():
    """nmf(..., M_global=M): weights, user callables and warm starts raise before anything touches a GPU or a process group"""
    import proxmin_amd as pm
    Y, A, S = np.ones((4, 6), np.float32), np.ones((4, 2), np.float32), np.ones((2, 6), np.float32)
    with pytest.raises(NotImplementedError):
        pm.nmf.nmf(Y, A, S, W=np.ones((4, 6)), M_global=8)
    with pytest.raises(NotImplementedError):
        pm.nmf.nmf(Y, A, S, step=lambda *X, it=None: (1.0, 1.0), M_global=8)
    with pytest.raises(NotImplementedError):
        pm.nmf.nmf(Y, A, S, callback=lambda *X, it=None: None, M_global=8)
    with pytest.raises(NotImplementedError):
        pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, M=(A, S), M_global=8)
    with pytest.raises(NotImplementedError):
        pm.nmf.nmf(Y, A, S, backtracking=True, f=lambda *X: 0.0, M_global=8)
