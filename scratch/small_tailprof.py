import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMX_TAIL_PROF"] = "1"
import numpy as np
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
Y, A0, S0 = orc.synthetic_problem(200, 1000, 5, np.float32, seed=1)
A, S = A0.copy(), S0.copy()
pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, max_iter=40, e_rel=1e-3, check_convergence=False)
