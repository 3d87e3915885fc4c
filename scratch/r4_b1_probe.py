# scratch: fuzz_nmf3.py 53 250, case 133 -- 132 x 407 x 2 adaprox / amsgrad with a decaying b1 ARRAY, 36 iterations, mode f32: 11 % of the entries up to 3.4 x the
# bound from the fp64 oracle while the fp32 oracle agrees with it everywhere.  Which ingredient?
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
import logging
logging.getLogger("proxmin").setLevel(logging.ERROR)
def frac(a, b):
    r = np.abs(a.astype(np.float64) - b) / (2e-5 + 2e-4 * np.abs(b))
    return float((r <= 1).mean()), float(r.max())
for M, N, K in ((132, 407, 2), (300, 400, 8)):
    for scheme in ("amsgrad", "adam"):
        for b1name, b1f in (("0.9", lambda n: 0.9), ("array", lambda n: 0.9 * 0.98 ** np.arange(n)), ("const array", lambda n: np.full(n, 0.9))):
            for its in (6, 36):
                Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=7)
                kw = dict(scheme=scheme, b1=b1f(its), b2=0.999, eps=1e-8, check_convergence=True, max_iter=its, e_rel=1e-6)
                A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
                orc.adaprox_nmf(Y.astype(np.float64), A64, S64, ("plus",), ("plus",), **kw)
                A32, S32 = A0.copy(), S0.copy()
                orc.adaprox_nmf(Y, A32, S32, ("plus",), ("plus",), **kw)
                A, S = A0.copy(), S0.copy()
                pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, **kw)
                print("%dx%dx%d %-7s b1 %-11s its %2d: fp32 oracle A %.4f/%.1f S %.4f/%.1f | device A %.4f/%.1f S %.4f/%.1f" % ((M, N, K, scheme, b1name, its) + frac(A32, A64) + frac(S32, S64) + frac(A, A64) + frac(S, S64)), flush=True)
