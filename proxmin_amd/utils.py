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
    It has no host implementation: calling `.step` directly on ndarrays is not supported."""

    def __init__(self, type=1, init_r=0.1):
        assert type in [1, 2]
        self.r = init_r
        self.type = type

    def step(self, *X, it=None, grads=None):
        raise NotImplementedError("BarzilaiBorweinStepper runs inside the device solver: pass it as step= to nmf(..., algorithm=pgm)")

    __call__ = step
