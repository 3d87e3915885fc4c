"""Constrained matrix factorisation front-end: same call signature and in-place contract as
`proxmin.nmf.nmf` (proxmin/nmf.py:96-203), executed on the MI355X.

    nmf(Y, A, S, W=1, prox_A=prox_plus, prox_S=prox_plus, algorithm=pgm, step=None,
        max_iter=1000, e_rel=1e-3, callback=None, **algorithm_args)

Y is uploaded once, A and S live on the device (S as S^T) for the whole call, and are written
back into the caller's arrays at exit (and before every user callback).  The helper functions
`log_likelihood`, `grad_likelihood`, `step_pgm`, `step_adaprox` keep the reference's signatures
and run the same device kernels on the arrays they are given.
"""
from __future__ import annotations

import atexit
import logging
import os
import threading
from functools import partial

import numpy as np

from . import algorithms, operators
from .engine import DeviceNMF, open_weighted, as_device_array, DeviceArrayRef

logger = logging.getLogger("proxmin")


_AMBIGUOUS = "The truth value of an array with more than one element is ambiguous. Use a.any() or a.all()"


def _weights(W, shape):
    """None for the reference's default W == 1, else the M x N weight array (a scalar is broadcast)."""
    if np.isscalar(W):
        if W == 1:
            return None
        return np.full(shape, W, dtype=np.float64)      # (fp32 contexts round it to float32 as before; fp64 ones keep the caller's number)
    W = np.asarray(W)
    assert W.shape == tuple(shape), "W must be M x N"
    return W


def _shape_of(Y):
    ref = as_device_array(Y)
    return ref.shape if ref is not None else np.shape(Y)


def _device_for(A, S, Y, W=None):
    """Context with Y (and W) and the factors on the device (weights: engine.open_weighted picks the kernel)."""
    A, S = np.asarray(A), np.asarray(S)
    Y = as_device_array(Y) or np.asarray(Y)      # a Y that lives in HBM is adopted in place (engine.DeviceArrayRef)
    device = getattr(Y, "device", 0) if isinstance(Y, DeviceArrayRef) else 0
    # [r6] float64 arrays: the fp64 kernels, like the reference's own arithmetic (nmf.py:39-41)
    from .engine import f64_applies
    if all(x.dtype == np.float64 for x in (Y, A, S)) and f64_applies(Y.shape[0], Y.shape[1], A.shape[1], weighted=W is not None):
        try:
            dev = DeviceNMF(Y.shape[0], Y.shape[1], A.shape[1], device=device, mode="f64" if W is None else "f64mfma")
        except NotImplementedError:
            dev = None
        if dev is not None:
            dev.set_Y(Y)
            if W is not None:
                dev.set_W(W)
            dev.set_factors(A, S)
            return dev
    if isinstance(Y, DeviceArrayRef) and Y.dtype == np.float64:
        raise NotImplementedError("a float64 device-resident Y goes with float64 factors (the fp64 kernels, K <= 128)")
    dev = open_weighted(Y.shape[0], Y.shape[1], A.shape[1], W, device=device)
    dev.set_Y(Y)
    dev.set_factors(A, S)
    return dev


_FACTOR_CTX = {}      # (device, M, N, K) -> the ONE factors-only context kept between calls
_FACTOR_LOCK = threading.RLock()


def _close_factor_ctx():
    """at interpreter exit, BEFORE libpmx can be unloaded under a context's __del__ (atexit runs ahead of module teardown)"""
    with _FACTOR_LOCK:
        for old in _FACTOR_CTX.values():
            old.close()
        _FACTOR_CTX.clear()


atexit.register(_close_factor_ctx)


class _Locked:
    """`with` wrapper around the cached factors-only context: holds the cache's lock from set_factors() to the result (two threads
    calling step_pgm / step_adaprox on the same shape used to race on the uploads and the status buffer: ADVICE r5) and leaves the
    context open."""

    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        return self.dev

    def __exit__(self, *exc):
        _FACTOR_LOCK.release()
        return False


def _factors_only(A, S, f64_ok=False):
    """Context holding just the factors (the step rules never touch Y: nothing M x N is allocated or uploaded).  [r5] The last one is
    kept: the reference's FISTA idiom `step=lambda *X, it=None: tuple(.5 * s for s in step_pgm(*X))` calls this once per iteration,
    and a context per call (allocations, a stream, two uploads) cost more than the rule itself.  One shape and device at a time; the
    result is a function of (A, S) alone -- pmx_step_pgm restarts its power iteration on a context without a solver.  [r6] The cache is
    guarded by a lock held for the whole call (the reference's step_pgm is a pure, thread-safe function), keyed by the device
    (PMX_DEVICE, default 0) as well, and closed by an atexit hook."""
    A, S = np.asarray(A), np.asarray(S)
    device = int(os.environ.get("PMX_DEVICE", "0"))
    # [r6] float64 factors: lambda_max from fp64 Gram matrices (the reference calls LAPACK on float64 arrays, utils.py:14-35)
    from .engine import f64_applies
    mode = "f64" if (f64_ok and A.dtype == np.float64 and S.dtype == np.float64 and f64_applies(A.shape[0], S.shape[1], A.shape[1])) else "f32"
    key = (device, A.shape[0], S.shape[1], A.shape[1], mode)
    _FACTOR_LOCK.acquire()
    try:
        dev = _FACTOR_CTX.get(key)
        if dev is None or getattr(dev, "h", None) is None:
            for old in _FACTOR_CTX.values():
                old.close()
            _FACTOR_CTX.clear()
            dev = _FACTOR_CTX[key] = DeviceNMF(key[1], key[2], key[3], device=device, mode=mode)
        dev.set_factors(A, S)
    except BaseException:
        _FACTOR_LOCK.release()
        raise
    return _Locked(dev)


def log_likelihood(*X, Y=0, W=1):
    """1/2 sum W (Y - A S)^2 (nmf.py:13-25), reduced inside the fused residual kernel."""
    A, S = X
    with _device_for(A, S, Y, _weights(W, _shape_of(Y))) as dev:
        return dev.loglike()


def grad_likelihood(*X, Y=0, W=1):
    """(D S^T, A^T D) with D = W (A S - Y) (nmf.py:28-41): one launch of the fused residual-gradient
    kernel.  Returns arrays in the dtype of A."""
    A, S = X
    with _device_for(A, S, Y, _weights(W, _shape_of(Y))) as dev:
        gA, gS = dev.grad()
    dt = np.asarray(A).dtype
    return gA.astype(dt), np.ascontiguousarray(gS).astype(dt)


def step_A(A, S):
    """1 / lmax(S S^T) (nmf.py:44-45)."""
    return step_pgm(A, S)[0]


def step_S(A, S):
    """1 / lmax(A^T A) (nmf.py:48-49)."""
    return step_pgm(A, S)[1]


def step_pgm(*X, it=None, W=1):
    """Lipschitz step sizes for PGM (nmf.py:52-65, W == 1 branch), Gram matrices and the largest
    eigenvalues computed on the device.  With an array W the reference's `if W == 1` raises ValueError (nmf.py:63); a
    scalar W != 1 reaches its sparse weighted rule, which is not implemented here."""
    if not np.isscalar(W):
        raise ValueError(_AMBIGUOUS)
    if W != 1:
        raise NotImplementedError("the weighted step rule of nmf.step_pgm (nmf.py:64-88) is not implemented; pass `step`")
    A, S = X
    with _factors_only(A, S, f64_ok=True) as dev:       # no Y: the rule needs the two K x K Gram matrices only
        return dev.step_pgm()


def step_adaprox(*X, it=None):
    """(mean(A, axis=0) / 10, mean(S, axis=1)[:, None] / 10) (nmf.py:91-93)."""
    A, S = X
    with _factors_only(A, S) as dev:
        aA, aS = dev.step_adaprox()
    dt = np.asarray(A).dtype
    return aA.astype(dt), aS.astype(dt)[:, None]


class scaled_step_pgm:
    """`step=scaled_step_pgm(c)`: the default Lipschitz rule multiplied by a constant, evaluated on
    the device each iteration.  Equivalent to the reference idiom
    `step=lambda *X, it=None: tuple(c * s for s in step_pgm(*X))` (needed e.g. for FISTA, which
    collapses with the undamped rule -- SURVEY.md section 4)."""

    def __init__(self, scale):
        self.scale = float(scale)

    def __call__(self, *X, it=None):
        return tuple(self.scale * s for s in step_pgm(*X))


class constant_step:
    """`step=constant_step(a_A, a_S)`: fixed step sizes, e.g. the adaprox learning rates of
    examples/unmixing.py:139-143 (`lambda *X, it: (alpha, alpha)`)."""

    def __init__(self, step_A, step_S=None):
        self.steps = (float(step_A), float(step_A if step_S is None else step_S))

    def __call__(self, *X, it=None, grads=None):
        return self.steps


def bsdmm_closures(Y, prox, W=1):
    """(proxs_f, steps_f_cb) for `algorithms.bsdmm(X, proxs_f, steps_f_cb, ...)`: the two closures the reference's nmf()
    builds inline (nmf.py:181-193) -- prox_f(X, step, Xs, j) = prox[j](X - step * grad(*Xs)[j], step) and
    step_f(Xs, j) = step_pgm(*Xs)[j].  Called directly they run the library's stand-alone device entry points; passed to
    algorithms.bsdmm they identify the problem (Y, the operators), and the whole solver runs as kernel chains."""
    grad = partial(grad_likelihood, Y=Y, W=W)
    prox = list(prox)

    def prox_f(X, step, Xs=None, j=None):
        return prox[j](X - step * grad(*Xs)[j], step)

    def step_f(Xs, j=None):
        return step_pgm(*Xs)[j]

    prox_f._pmx_nmf = step_f._pmx_nmf = (grad, prox)
    return prox_f, step_f


def _nmf_sharded(Y, A, S, W, prox_A, prox_S, algorithm, step, max_iter, e_rel, callback, kw):
    """nmf(Y_local, A_local, S, ..., M_global=M[, group=pg, comm="torch"|"native", s_split="auto"]) -> the sharded drivers.
    What the sharded protocol does not carry raises here instead of silently running something else: weights, user callables (prox, step,
    callback: they would see one rank's rows), Barzilai-Borwein / backtracking.  Returns the reference's shapes with None where a
    gathered array would be needed (pgm's gradient, adaprox's moments): (converged, None, None) / (converged, None, None, None) / converged."""
    from . import distributed
    M_global = int(kw.pop("M_global"))
    group, comm, s_split = kw.pop("group", None), kw.pop("comm", None), kw.pop("s_split", "auto")
    if not (np.isscalar(W) and W == 1):
        raise NotImplementedError("row-sharded nmf(): a weighted likelihood is not carried by the sharded protocol")
    if callback is not None and not isinstance(callback, utils_NullCallback()):
        raise NotImplementedError("row-sharded nmf(): a callback would see one rank's rows of A only")
    assert np.asarray(A).shape[0] <= M_global
    if algorithm is algorithms.pgm:
        scale, fixed = 1.0, None
        if isinstance(step, scaled_step_pgm):
            scale = step.scale
        elif isinstance(step, constant_step):
            fixed = step.steps
        elif not (step is None or step is step_pgm or (isinstance(step, partial) and step.func is step_pgm)):
            raise NotImplementedError("row-sharded pgm: the Lipschitz rule (default, scaled_step_pgm) or constant_step only")
        if kw.pop("backtracking", False):
            raise NotImplementedError("row-sharded pgm: no line search")
        acc = bool(kw.pop("accelerated", False))
        if kw:
            raise TypeError("unexpected arguments for row-sharded pgm: %r" % sorted(kw))
        conv, _ = distributed.nmf_pgm_sharded(Y, A, S, M_global, prox_A=prox_A, prox_S=prox_S, accelerated=acc, step_scale=scale, fixed_steps=fixed,
                                              e_rel=e_rel, max_iter=max_iter, group=group, comm=comm, s_split=s_split)
        return conv, None, None
    if algorithm is algorithms.adaprox:
        if not (step is None or step is step_adaprox):
            raise NotImplementedError("row-sharded adaprox: the default step rule (nmf.step_adaprox) only")
        for bad in ("M", "V", "Vhat"):
            if kw.pop(bad, None) is not None:
                raise NotImplementedError("row-sharded adaprox: no warm start of the moments")
        conv, _ = distributed.nmf_adaprox_sharded(Y, A, S, M_global, prox_A=prox_A, prox_S=prox_S, e_rel=e_rel, max_iter=max_iter, group=group, comm=comm,
                                                  s_split=s_split, **kw)
        return conv, None, None, None
    if step is not None:
        raise NotImplementedError("a user `step` is not supported with bsdmm (it crashes in the reference too)")
    for bad in ("Ls", "update_order", "steps_g", "steps_g_update"):
        if kw.get(bad) is not None and bad != "steps_g_update":
            raise NotImplementedError("row-sharded bsdmm: %s is not carried by the sharded protocol" % bad)
        kw.pop(bad, None)
    conv, _ = distributed.nmf_bsdmm_sharded(Y, A, S, M_global, prox_A=prox_A, prox_S=prox_S, proxs_g=kw.pop("proxs_g", None), e_rel=e_rel,
                                            e_abs=kw.pop("e_abs", 0.0), max_iter=max_iter, group=group, comm=comm)
    if kw:
        raise TypeError("unexpected arguments for row-sharded bsdmm: %r" % sorted(kw))
    return conv


def utils_NullCallback():
    from .utils import NullCallback
    return NullCallback


def nmf(
    Y,
    A,
    S,
    W=1,
    prox_A=operators.prox_plus,
    prox_S=operators.prox_plus,
    algorithm=algorithms.pgm,
    step=None,
    max_iter=1000,
    e_rel=1e-3,
    callback=None,
    **algorithm_args
):
    """Non-negative / constrained matrix factorisation  minimise || Y - A S ||^2  (nmf.py:96-203).

    Args and returns are those of the reference: A (M x K) and S (K x N) are updated in place; the
    return value is that of the chosen algorithm (pgm: (converged, grads, steps); adaprox:
    (converged, M, V, Vhat); bsdmm: converged).  `algorithm` must be one of THIS package's
    `algorithms.pgm / adaprox / bsdmm` (identity test, nmf.py:141).

    Beyond the reference: Y may already live in HBM (anything with `__cuda_array_interface__`, float32: adopted in place); and with
    `M_global=<rows of the whole problem>` (plus optional `group=`, `comm=`, `s_split=`) the arrays are one rank's rows of a row-sharded
    run over a torch.distributed process group (see _nmf_sharded).
    """
    assert algorithm in [algorithms.pgm, algorithms.adaprox, algorithms.bsdmm]

    # [r6] Row-sharded runs through the SAME entry point: with `M_global=<rows of the whole problem>` the arrays are THIS RANK'S rows of Y
    # and A (S is replicated) and the call runs on the process group `group` (default: the world) with one packed collective per
    # iteration (proxmin_amd/distributed.py).  Without M_global the call is the reference's: one process, one GPU.
    if "M_global" in algorithm_args:
        return _nmf_sharded(Y, A, S, W, prox_A, prox_S, algorithm, step, max_iter, e_rel, callback, dict(algorithm_args))

    grad = partial(grad_likelihood, Y=Y, W=W)
    X = [A, S]
    prox = [prox_A, prox_S]

    if algorithm is algorithms.pgm:
        if step is None:
            step = partial(step_pgm, W=W)
        return algorithm(X, grad, step, prox=prox, max_iter=max_iter, e_rel=e_rel, callback=callback, **algorithm_args)

    if algorithm is algorithms.adaprox:
        if step is None:
            step = step_adaprox
        return algorithm(X, grad, step, prox=prox, max_iter=max_iter, e_rel=e_rel, callback=callback, **algorithm_args)

    if algorithm is algorithms.bsdmm:
        if step is not None:
            # the reference raises UnboundLocalError here (nmf.py:187-198 passes the undefined step_f)
            raise NotImplementedError("a user `step` is not supported with bsdmm (it crashes in the reference too)")
        prox_f, step_f = bsdmm_closures(Y, prox, W=W)
        return algorithm(X, prox_f, step_f, max_iter=max_iter, e_rel=e_rel, callback=callback, **algorithm_args)
