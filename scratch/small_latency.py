# scratch: end-to-end latency of nmf() on small problems (context set-up, uploads, chain, download)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
for (M, N, K, its) in ((200, 1000, 5, 100), (200, 1000, 5, 1000), (2000, 2000, 16, 100)):
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=1)
    for algo in ("pgm", "adaprox"):
        ts = []
        for rep in range(4):
            A, S = A0.copy(), S0.copy()
            t0 = time.perf_counter()
            if algo == "pgm":
                pm.nmf.nmf(Y, A, S, max_iter=its, e_rel=1e-12)
            else:
                pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, max_iter=its, e_rel=1e-3, check_convergence=False)
            ts.append(time.perf_counter() - t0)
        Ao, So = A0.copy(), S0.copy()
        t0 = time.perf_counter()
        (orc.pgm_nmf(Y, Ao, So, max_iter=its, e_rel=1e-12) if algo == "pgm" else orc.adaprox_nmf(Y, Ao, So, max_iter=its, e_rel=1e-3, check_convergence=False))
        tc = time.perf_counter() - t0
        print("%dx%dx%d %s %d its: device %.1f ms (first call %.1f), oracle on CPU %.1f ms" % (M, N, K, algo, its, 1e3 * min(ts[1:]), 1e3 * ts[0], 1e3 * tc), flush=True)
