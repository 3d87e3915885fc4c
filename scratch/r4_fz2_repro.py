import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from functools import partial
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
ops = pm.operators
Y, A0, S0 = orc.synthetic_problem(300, 200, 8, np.float32, seed=1)
W = (0.1 + np.random.default_rng(0).random((300, 200))).astype(np.float32)
for name, kw in (("weighted pgm scaled step", dict(W=W, step=pm.nmf.scaled_step_pgm(1.0))),
                 ("weighted fista scaled step", dict(W=W, step=pm.nmf.scaled_step_pgm(0.5), accelerated=True)),
                 ("bt", dict(backtracking=True)),
                 ("bt hard_plus", dict(backtracking=True, prox_A=partial(ops.prox_hard_plus, thresh=0.01, type="absolute"))),
                 ("bt accel scaled", dict(backtracking=True, accelerated=True, step=pm.nmf.scaled_step_pgm(0.5))),
                 ("bt weighted scaled", dict(backtracking=True, W=W, step=pm.nmf.scaled_step_pgm(1.0)))):
    A, S = A0.copy(), S0.copy()
    try:
        pm.nmf.nmf(Y, A, S, max_iter=3, e_rel=1e-12, **kw)
        print("OK  ", name)
    except Exception:
        print("EXC ", name)
        traceback.print_exc()
