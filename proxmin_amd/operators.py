"""Proximal operators of the nmf() path -- same names, arguments and in-place behaviour as
proxmin/operators.py:20-160 -- backed by the HIP operator kernel (csrc/k_update.hip, prox_one).

Two uses:
  * passed to `nmf.nmf()` as `prox_A` / `prox_S` / `proxs_g` (bare, or wrapped in
    `functools.partial` for `axis=` / `thresh=` / `type=`, or combined with `AlternatingProjections`),
    they are recognised by identity -- the same test as AlternatingProjections.find,
    operators.py:213-224 -- and run FUSED inside the solver kernels; the Python bodies below are
    never called on that path;
  * called directly on an ndarray they run the same device kernel on that array
    (`pmx_prox_array`): there is no NumPy implementation in this package.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from . import _lib

__all__ = ["prox_id", "prox_zero", "prox_plus", "prox_unity", "prox_unity_plus", "prox_min", "prox_max",
           "prox_hard", "prox_hard_plus", "prox_soft", "prox_soft_plus", "AlternatingProjections"]


# ---------------------------------------------------------------------------------------------
# direct application on host arrays
# ---------------------------------------------------------------------------------------------
def _seq(entries, repeat=1):
    ps = _lib.ProxSeq()
    ps.n = len(entries)
    ps.repeat = repeat
    for i, (op, unit, thresh, relative) in enumerate(entries):
        ps.seq[i].op, ps.seq[i].unit, ps.seq[i].thresh, ps.seq[i].relative = _lib.PROX[op], unit, float(thresh), int(relative)
    return ps


def _run_rows(rows2d, ps, step_k):
    lib = _lib.require_gpu()
    buf = np.ascontiguousarray(rows2d, dtype=np.float32)
    sk = np.ascontiguousarray(np.broadcast_to(np.asarray(step_k, dtype=np.float32).reshape(-1), (buf.shape[1],)))
    _lib.check(lib.pmx_prox_array(0, buf.ctypes.data_as(C.c_void_p), buf.shape[0], buf.shape[1], C.byref(ps),
                                  sk.ctypes.data_as(C.c_void_p)))
    return buf


def _apply(X, step, op, axis=None, thresh=0.0, kind="relative"):
    """Run one operator on the ndarray X in place (any float dtype; computed in float32)."""
    assert kind in ["relative", "absolute"]
    X = np.asarray(X) if not isinstance(X, np.ndarray) else X
    rel = kind == "relative"
    step_arr = np.asarray(step, dtype=np.float64)
    if op in ("unity", "unity_plus"):
        if X.ndim != 2 or axis not in (0, 1):
            raise NotImplementedError("prox_unity on the device needs a 2-D array and axis in (0, 1)")
        V = X if axis == 1 else X.T
        if V.shape[1] <= _lib.MAXK:
            out = _run_rows(V, _seq([(op, 0, 0.0, 0)]), 0.0)
            X[...] = out if axis == 1 else out.T
            return X
        # long normalised axis: sums over the ROWS of the device array (column sums folded across the grid), the other axis
        # in groups of at most MAXK independent columns
        Wm = X if axis == 0 else X.T
        for c0 in range(0, Wm.shape[1], _lib.MAXK):
            Wm[:, c0:c0 + _lib.MAXK] = _run_rows(np.ascontiguousarray(Wm[:, c0:c0 + _lib.MAXK]), _seq([(op, 1, 0.0, 0)]), 0.0)
        return X
    ps = _seq([(op, 0, thresh, rel)])
    if step_arr.ndim == 0 or step_arr.size == 1 or not rel or op in ("id", "zero", "plus"):
        flat = X.reshape(-1)
        W = _lib.MAXK
        pad = (-flat.size) % W
        buf = np.concatenate([flat.astype(np.float32), np.zeros(pad, np.float32)]).reshape(-1, W)
        out = _run_rows(buf, ps, float(step_arr.reshape(-1)[0]) if step_arr.size else 0.0)
        X[...] = out.reshape(-1)[: flat.size].reshape(X.shape)
        return X
    # per-component step (what adaprox passes: shape (K,) for A, (K,1) for S -- nmf.py:93)
    if X.ndim == 2 and step_arr.shape in ((X.shape[1],), (1, X.shape[1])) and X.shape[1] <= _lib.MAXK:
        X[...] = _run_rows(X, ps, step_arr.reshape(-1))
        return X
    if X.ndim == 2 and step_arr.shape == (X.shape[0], 1) and X.shape[0] <= _lib.MAXK:
        X[...] = _run_rows(X.T, ps, step_arr.reshape(-1)).T
        return X
    raise NotImplementedError("step of shape %s does not broadcast per component over X of shape %s" % (step_arr.shape, X.shape))


def prox_id(X, step):
    """Identity proximal operator (operators.py:20-23)."""
    return X


def prox_zero(X, step):
    """Projection onto zero (operators.py:26-30)."""
    return _apply(X, step, "zero")


def prox_plus(X, step):
    """Projection onto non-negative numbers (operators.py:33-38)."""
    return _apply(X, step, "plus")


def prox_unity(X, step, axis=0):
    """Projection onto sum=1 along an axis (operators.py:41-45)."""
    return _apply(X, step, "unity", axis=axis)


def prox_unity_plus(X, step, axis=0):
    """Non-negative projection onto sum=1 along an axis (operators.py:48-52)."""
    return _apply(X, step, "unity_plus", axis=axis)


def prox_min(X, step, thresh=0, type="relative"):
    """Projection onto numbers above `thresh` (operators.py:55-68)."""
    return _apply(X, step, "min", thresh=thresh, kind=type)


def prox_max(X, step, thresh=0, type="relative"):
    """Projection onto numbers below `thresh` (operators.py:71-84)."""
    return _apply(X, step, "max", thresh=thresh, kind=type)


def prox_hard(X, step, thresh=0, type="relative"):
    """Hard thresholding (operators.py:109-124)."""
    return _apply(X, step, "hard", thresh=thresh, kind=type)


def prox_hard_plus(X, step, thresh=0, type="relative"):
    """Hard thresholding with projection onto non-negative numbers (operators.py:127-135)."""
    return _apply(X, step, "hard_plus", thresh=thresh, kind=type)


def prox_soft(X, step, thresh=0, type="relative"):
    """Soft thresholding (operators.py:138-150)."""
    return _apply(X, step, "soft", thresh=thresh, kind=type)


def prox_soft_plus(X, step, thresh=0, type="relative"):
    """Soft thresholding with projection onto non-negative numbers (operators.py:153-160)."""
    return _apply(X, step, "soft_plus", thresh=thresh, kind=type)


class AlternatingProjections(object):
    """POCS composition of several operators (operators.py:187-224): the list is applied
    last-to-first, `repeat` times.  Inside nmf() a list of built-ins becomes one fused device
    operator sequence; called directly it applies the members one after the other."""

    def __init__(self, prox_list=None, repeat=1):
        self.operators = []
        self.repeat = repeat
        if prox_list is not None:
            self.operators += prox_list

    def __call__(self, X, step):
        for _ in range(self.repeat):
            for prox in self.operators[::-1]:
                X = prox(X, step)
        return X

    def find(self, cls):
        for i, prox in enumerate(self.operators):
            if isinstance(prox, functools.partial):
                if prox.func is cls:
                    return i
            elif prox is cls:
                return i
        return -1


# ---------------------------------------------------------------------------------------------
# recognition of built-ins (used by nmf(): callable -> device operator sequence)
# ---------------------------------------------------------------------------------------------
_BY_FUNC = {prox_id: "id", prox_zero: "zero", prox_plus: "plus", prox_unity: "unity", prox_unity_plus: "unity_plus",
            prox_min: "min", prox_max: "max", prox_hard: "hard", prox_hard_plus: "hard_plus",
            prox_soft: "soft", prox_soft_plus: "soft_plus"}


def _one_entry(prox, block):
    func, kw = prox, {}
    if isinstance(prox, functools.partial):
        if prox.args:
            return None
        func, kw = prox.func, dict(prox.keywords)
    op = _BY_FUNC.get(func)
    if op is None:
        return None
    if op in ("unity", "unity_plus"):
        axis = kw.pop("axis", 0)
        if kw or axis not in (0, 1):
            return None
        # device layout: A is M x K, S is held as S^T (N x K).  unit 0 = along the K components.
        unit = 0 if (block == 0 and axis == 1) or (block == 1 and axis == 0) else 1
        return (op, unit, 0.0, 0)
    if op in ("id", "zero", "plus"):
        return None if kw else (op, 0, 0.0, 0)
    thresh = kw.pop("thresh", 0)
    kind = kw.pop("type", "relative")
    if kw:
        return None
    assert kind in ["relative", "absolute"]
    return (op, 0, float(thresh), kind == "relative")


class NotFusable(NotImplementedError):
    """An operator of this module that the fused solver kernels do not contain (prox_unity* along the long axis: a
    grid-wide sum per application).  The solvers then apply it between kernel launches by calling it on the host copy of
    its argument -- which runs the stand-alone device operator kernel -- one iteration per call."""


def device_proxseq(prox, block, for_solver=False):
    """Translate a prox callable into a device operator sequence for factor `block` (0 = A, 1 = S).
    Returns a _lib.ProxSeq (n == 0 for prox=None) or raises NotImplementedError for callables that
    are not (compositions of) this module's operators.  for_solver: also raise (NotFusable) for operators that exist
    only as stand-alone kernels."""
    if for_solver and prox is not None:
        seq = device_proxseq(prox, block)
        if any(seq.seq[i].unit != 0 for i in range(seq.n)):
            raise NotFusable("prox_unity along the long axis of block %d is applied between kernel launches" % block)
        return seq
    if prox is None:
        return _seq([])
    if isinstance(prox, AlternatingProjections):
        entries = [_one_entry(q, block) for q in prox.operators[::-1]]
        if any(e is None for e in entries) or len(entries) > _lib.MAX_SEQ:
            raise NotImplementedError("AlternatingProjections of user-defined operators (or more than %d) cannot run on the device" % _lib.MAX_SEQ)
        return _seq(entries, repeat=int(prox.repeat))
    e = _one_entry(prox, block)
    if e is None:
        raise NotImplementedError(
            "prox %r is not one of proxmin_amd.operators (bare, functools.partial or AlternatingProjections): "
            "user-defined Python prox callables are not supported on the device path" % (prox,))
    return _seq([e])
