"""small problems: per-iteration time of adaprox / bsdmm / pgm with fp32 inputs (fused tail with grid barriers) and fp64 inputs (single-workgroup fp64 kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import numpy as np
from functools import partial
import proxmin_amd as pm
from oracle import nmf_oracle as orc
import logging
logging.getLogger("proxmin").setLevel(logging.ERROR)
for M, N, K in ((200, 1000, 5), (33, 47, 3), (1000, 1000, 8), (4000, 250, 16)):
    for dt in (np.float32, np.float64):
        Y, A0, S0 = orc.synthetic_problem(M, N, K, dt, unity_S=True, seed=3)
        for algo in ("pgm", "adaprox", "adaprox_unity", "bsdmm"):
            ts = []
            for its in (50, 1050):
                A, S = A0.copy(), S0.copy()
                t0 = time.perf_counter()
                if algo == "pgm":
                    pm.nmf.nmf(Y, A, S, max_iter=its, e_rel=1e-14)
                elif algo == "bsdmm":
                    pg = [[pm.operators.prox_plus, partial(pm.operators.prox_soft, thresh=1e-3)]] * 2
                    pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=pg, max_iter=its, e_rel=1e-14)
                else:
                    pS = partial(pm.operators.prox_unity_plus, axis=0) if algo.endswith("unity") else pm.operators.prox_plus
                    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="amsgrad", prox_S=pS, max_iter=its, e_rel=1e-3, check_convergence=False)
                ts.append(time.perf_counter() - t0)
            print("%5d x %5d x %2d %-8s %-14s %.1f us / iteration" % (M, N, K, np.dtype(dt).name, algo, (ts[1] - ts[0]) / 1000 * 1e6), flush=True)
