"""proxmin_amd -- the proxmin NMF/CMF hot path (`nmf.nmf` + pgm / adaprox / bsdmm + prox operators)
running on AMD Instinct MI355X (gfx950) through hand-written HIP kernels.

Drop-in for the reference's `proxmin.nmf.nmf()` path:

    import proxmin_amd as proxmin
    from proxmin_amd import nmf, operators
    nmf.nmf(Y, A, S, prox_A=operators.prox_plus, algorithm=proxmin.algorithms.adaprox, scheme="amsgrad")

Mirrors proxmin/__init__.py:1-4 (star-exports of algorithms and operators, submodules nmf, utils).
"""
from .algorithms import pgm, adaprox, bsdmm  # noqa: F401
from .operators import *  # noqa: F401,F403
from . import algorithms, operators, nmf, utils  # noqa: F401
from .engine import set_default_mode, get_default_mode, LIBRARY_DEFAULT_MODE  # noqa: F401

__version__ = "0.1.0"
