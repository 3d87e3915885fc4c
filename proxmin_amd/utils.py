"""Host-side helpers of the nmf() path that are plain Python in the reference and stay plain Python
here (callback plumbing and the scalar Nesterov sequence); everything array-sized lives on the GPU.
"""
from __future__ import annotations



def _as_tuple(X):
    """proxmin/utils.py:8-12."""
    return X if type(X) in (list, tuple) else (X,)


class Traceback(object):
    """Callback that stores a copy of every iterate it is shown (proxmin/utils.py:104-116)."""

    def __init__(self):
        self._trace = []

    def __call__(self, *X, it=None):
        self._trace.append(tuple(x.copy() for x in X))

    @property
    def trace(self):
        return self._trace

    def clear(self):
        self._trace = []


def l2sq(x):
    """sum of squares (proxmin/utils.py:257-260) -- used by the host-side proximal loop around a user-defined prox"""
    return (x ** 2).sum()


class NullCallback(object):
    """proxmin/utils.py:119-121.  nmf() treats it like callback=None: no per-iteration D2H copies."""

    def __call__(self, *X, it):
        pass


class BarzilaiBorweinStepper:
    """Barzilai-Borwein step sizes with Burdakov stabilisation (proxmin/utils.py:209-241).

    Pass the object (or its bound `.step`) as `step=` to `nmf(..., algorithm=pgm)`: the reductions
    (sum s^2, s.y, y^2, |G|^2, max|X|, max|G| with s = X - X_prev, y = G - G_prev) and the step formula run
    on the device each iteration, evaluated -- like the reference -- at the point the gradient was taken.
    [r5] Called directly on ndarrays -- `stepper.step(*X, it=it, grads=G)`, the reference's own signature and state
    (`X_`, `G_`, `Delta`) -- the six sums of every block come from the device (pmx_bb_sums: the arrays go up, six numbers
    come back); the scalar formula on them is the reference's.  Return types are the reference's: a tuple of
    scalars at it == 0, an ndarray of N steps afterwards (utils.py:222,241)."""

    device = None       # GPU the stand-alone sums run on: None = PMX_DEVICE from the environment, else device 0 (set the attribute to choose)

    def __init__(self, type=1, init_r=0.1):
        assert type in [1, 2]
        self.r = init_r
        self.type = type

    def _sums(self, x, g, xp, gp):
        import ctypes as C
        import os
        import numpy as np
        from . import _lib
        lib = _lib.require_gpu()
        device = int(os.environ.get("PMX_DEVICE", "0")) if self.device is None else int(self.device)
        f64 = all(np.asarray(a).dtype == np.float64 for a in (x, g))
        dt = np.float64 if f64 else np.float32
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=dt) for a in (x, xp, g, gp)]
        assert arrs[0].size == arrs[2].size and (arrs[1] is None or arrs[1].size == arrs[0].size == arrs[3].size)
        out = (C.c_double * 6)()
        ptr = [None if a is None else a.ctypes.data_as(C.c_void_p) for a in arrs]
        _lib.check(lib.pmx_bb_sums(device, int(f64), ptr[0], ptr[1], ptr[2], ptr[3], arrs[0].size, out))
        return list(out)

    def step(self, *X, it=None, grads=None):
        import numpy as np
        N = len(X)
        assert grads is not None and len(grads) == N, "BarzilaiBorweinStepper.step needs grads= (algorithms.py:73-77 passes them)"
        first = it == 0
        sums = [self._sums(X[j], grads[j], None if first else self.X_[j], None if first else self.G_[j]) for j in range(N)]
        self.X_ = tuple(np.array(x, copy=True) for x in X)       # utils.py:221,229 (_copy_tuple)
        self.G_ = grads                                           # "no copy needed, created fresh every single iteration"
        if first:
            self.Delta = np.array([np.inf, ] * N)
            return tuple(self.r * s[4] / s[5] for s in sums)
        with np.errstate(divide="ignore", invalid="ignore"):
            if self.type == 1:
                A = tuple(np.float64(s[0]) / np.float64(s[1]) for s in sums)
            else:
                A = tuple(np.float64(s[1]) / np.float64(s[2]) for s in sums)
            if it <= 3:
                self.Delta = np.minimum(self.Delta, tuple(np.sqrt(s[0]) for s in sums))
            Astab = tuple(self.Delta[j] / np.sqrt(np.float64(sums[j][3])) for j in range(N))
            return np.minimum(np.abs(A), Astab)

    __call__ = step
