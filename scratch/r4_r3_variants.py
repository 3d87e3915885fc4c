import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd import engine as eng
from oracle import nmf_oracle as orc
for M, N, K in ((2048, 2048, 64), (4096, 4096, 64)):
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=4321)
    r64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    line = "%d x %d x %d gA / gS rms error of max|g|:" % (M, N, K)
    for name, mode, r3 in (("f32", "f32", None), ("f16x2", "f16x2", "0"), ("2 accumulators, 3 products", "f16x2", "2"), ("<R3>", "f16x2", "1")):
        if r3 is not None: os.environ["PMX_F16_R3"] = r3
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A, S)
            gr = dev.grad()
        os.environ.pop("PMX_F16_R3", None)
        e = [float(np.sqrt(((gr[j] - r64[j]) ** 2).mean()) / np.abs(r64[j]).max()) for j in range(2)]
        line += "  %s %.2e / %.2e |" % (name, e[0], e[1])
    print(line, flush=True)
