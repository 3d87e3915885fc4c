# scratch: the two-term fp16 kernel against fp64 gradients, beside the exact-fp32 and split-bf16 kernels
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd.engine import DeviceNMF
from oracle import nmf_oracle as orc
for (M, N, K, near) in ((1536, 2048, 64, True), (1024, 768, 64, False), (2304, 4096, 64, False)):
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=31)
    if near:
        orc.adaprox_nmf(Y, A, S, scheme="amsgrad", max_iter=60, e_rel=1e-3, check_convergence=False)
    g64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    l64 = orc.half_sq_residual(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    for mode in ("f32", "bf16x3", "f16x2"):
        with DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A, S)
            gg = dev.grad(); loss = dev.loglike()
        err = [np.linalg.norm(gg[j] - g64[j]) / np.linalg.norm(g64[j]) for j in range(2)]
        mx = [np.abs(gg[j] - g64[j]).max() / np.abs(g64[j]).max() for j in range(2)]
        print("%dx%d near=%d %-7s relF gA %.2e gS %.2e | max/maxabs %.2e %.2e | loss rel %.1e" % (M, N, near, mode, err[0], err[1], mx[0], mx[1], abs(loss - l64) / l64), flush=True)
