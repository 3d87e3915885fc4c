# scratch: the one case fuzz_nmf3.py flagged (seed 22, case 7 of the first stream): nadam, b1 decaying, unity on S, 32 iterations, small problem, mode f32 --
# device against the fp64 oracle beside the oracle's own fp32 run (the yardstick), a few data seeds and iteration counts
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from functools import partial
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
import logging
logging.getLogger("proxmin").setLevel(logging.ERROR)
M, N, K = 467, 516, 8
def frac(a, b):
    return float((np.abs(a.astype(np.float64) - b) <= 2e-5 + 2e-4 * np.abs(b)).mean())
for sd in (1, 2, 3):
    for its in (8, 16, 32):
        Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=sd)
        kw = dict(scheme="nadam", b1=0.9 * 0.98 ** np.arange(its), b2=0.999, eps=1e-6, check_convergence=False, max_iter=its, e_rel=1e-3)
        A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
        orc.adaprox_nmf(Y.astype(np.float64), A64, S64, ("plus",), ("unity_plus", 0), **kw)
        A32, S32 = A0.copy(), S0.copy()
        orc.adaprox_nmf(Y, A32, S32, ("plus",), ("unity_plus", 0), **kw)
        line = "seed %d its %2d: yardstick (oracle fp32) A %.5f S %.5f |" % (sd, its, frac(A32, A64), frac(S32, S64))
        for mode in ("f32", "f16x2"):
            pm.set_default_mode(mode)
            A, S = A0.copy(), S0.copy()
            pm.nmf.nmf(Y, A, S, prox_S=partial(pm.operators.prox_unity_plus, axis=0), algorithm=pm.adaprox, **kw)
            line += " device %s A %.5f S %.5f |" % (mode, frac(A, A64), frac(S, S64))
        print(line, flush=True)
pm.set_default_mode("f32")
