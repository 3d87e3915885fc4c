# scratch: RAdam's first iterations (plain momentum steps of size alpha: factors of huge dynamic range) in mode f16x2, framed and unframed
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
import logging
logging.getLogger("proxmin").setLevel(logging.ERROR)
CASES = [(1408, 1024, 128), (1432, 1036, 128), (1408, 1024, 97), (1432, 1036, 97), (1024, 1024, 64), (1024, 1024, 50), (1024, 1024, 32), (1024, 1024, 20), (1433, 704, 119)]
for M, N, K in CASES:
    for scheme in ("radam", "adam"):
        for its in (1, 2, 3):
            Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=5)
            Ao, So = A0.astype(np.float64), S0.astype(np.float64)
            orc.adaprox_nmf(Y.astype(np.float64), Ao, So, ("plus",), ("plus",), scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False)
            line = "%5d x %5d x %3d %-5s its=%d max|A| %.3g max|S| %.3g:" % (M, N, K, scheme, its, np.abs(Ao).max(), np.abs(So).max())
            for mode in ("f32", "f16x2"):
                pm.set_default_mode(mode)
                A, S = A0.copy(), S0.copy()
                pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False)
                worst, frac, nan = 0.0, 1.0, 0
                for a, b in ((A, Ao), (S, So)):
                    nan += int(np.isnan(a).sum())
                    r = np.abs(a.astype(np.float64) - b) / (2e-5 + 2e-4 * np.abs(b))
                    worst = max(worst, float(np.nanmax(r))); frac = min(frac, float((r <= 1).mean()))
                line += "  %s worst %.1f frac %.5f nan %d" % (mode, worst, frac, nan)
            print(line, flush=True)
