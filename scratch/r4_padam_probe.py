# scratch: fuzz_nmf2.py 31 150 135 -- 835 x 2552 x 64 adaprox / padam, prox_A = prox_max(5, absolute), prox_S = prox_soft: the device returns NaN everywhere
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
ops = pm.operators
def nan_cb(tag):
    def cb(*X, it=None):
        print("   %s it %d: NaN in A %d, S %d; max|A| %.3g max|S| %.3g" % (tag, it, int(np.isnan(X[0]).sum()), int(np.isnan(X[1]).sum()), np.nanmax(np.abs(X[0])), np.nanmax(np.abs(X[1]))), flush=True)
    return cb
for M, N, K in ((835, 2552, 64), (896, 2560, 64), (300, 400, 8)):
    for scheme in ("padam", "adam"):
        for pA, sA, pS, sS, nm in ((partial(ops.prox_max, thresh=5.0, type="absolute"), ("max", 5.0, "absolute"), partial(ops.prox_soft, thresh=1e-2), ("soft", 1e-2, "relative"), "max/soft"),
                                   (ops.prox_plus, ("plus",), partial(ops.prox_soft, thresh=1e-2), ("soft", 1e-2, "relative"), "plus/soft"),
                                   (partial(ops.prox_max, thresh=5.0, type="absolute"), ("max", 5.0, "absolute"), ops.prox_plus, ("plus",), "max/plus")):
            Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=3)
            A, S = A0.copy(), S0.copy()
            pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.adaprox, scheme=scheme, max_iter=5, e_rel=1e-3, check_convergence=False)
            Ao, So = A0.astype(np.float64), S0.astype(np.float64)
            orc.adaprox_nmf(Y.astype(np.float64), Ao, So, sA, sS, scheme=scheme, max_iter=5, e_rel=1e-3, check_convergence=False)
            print("%dx%dx%d %s %s: device NaN A %d S %d | oracle NaN A %d S %d | oracle max|A| %.3g max|S| %.3g" % (M, N, K, scheme, nm, int(np.isnan(A).sum()), int(np.isnan(S).sum()),
                  int(np.isnan(Ao).sum()), int(np.isnan(So).sum()), np.nanmax(np.abs(Ao)), np.nanmax(np.abs(So))), flush=True)
