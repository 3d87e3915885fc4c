# scratch: the Gram-matrix correction of the two-term fp16 gradients (k_grad_fix, PMX_F16_GRAMFIX=1): gradient error against fp64 beside plain f16x2, <R3> and exact fp32
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd import engine as eng
from oracle import nmf_oracle as orc
for M, N, K in ((2048, 2048, 64), (2048, 2048, 32), (2048, 2048, 128), (1000, 1500, 50)):
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=4321)
    r64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    line = "%d x %d x %d  gA / gS rms error of max|g|:" % (M, N, K)
    for name, mode, env in (("f32", "f32", {}), ("f16x2", "f16x2", {}), ("f16x2+fix", "f16x2", {"PMX_F16_GRAMFIX": "1"}), ("f16x2r", "f16x2r", {})):
        os.environ.update(env)
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A, S)
            gr = dev.grad(); g2 = dev.grad()
            assert np.array_equal(gr[0], g2[0]) and np.array_equal(gr[1], g2[1])
            kern = dev.k1_info()["kernel"]
        for k in env: del os.environ[k]
        e = [float(np.sqrt(((gr[j] - r64[j]) ** 2).mean()) / np.abs(r64[j]).max()) for j in range(2)]
        line += "  %s (%s) %.2e / %.2e |" % (name, kern.replace("k_grad_", ""), e[0], e[1])
    print(line, flush=True)
