# scratch: the PCIe-inclusive rate -- proxmin_amd.nmf.nmf() called with HOST arrays at cfg3's size (Y 1 GiB fp32 uploaded inside the call, A / S written back at exit)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from functools import partial
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
M = N = 16384; K = 64
Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=1234)
pm.set_default_mode("f16x2r")
for its in (1, 50, 200):
    A, S = A0.copy(), S0.copy()
    t0 = time.perf_counter()
    pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(pm.operators.prox_unity_plus, axis=0), max_iter=its, e_rel=1e-3, check_convergence=False)
    dt = time.perf_counter() - t0
    print("nmf() with host arrays, %3d iterations: %.1f ms end to end = %.1f it/s (context + 1 GiB upload of Y + iterations + write-back of A, S)" % (its, 1e3 * dt, its / dt), flush=True)
