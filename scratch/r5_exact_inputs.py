# scratch [r5]: where does the fp16 modes' coherent gradient bias come from?  Inputs whose every intermediate is EXACT in the split arithmetic (small integers: A, S in
# {0..3}, Y integer -> P, R integers < 2048: one fp16 term holds them): what is left is the fp32 ACCUMULATION of the gradient contractions.  Bias and rms against exact.
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd import engine as eng
M, N, K = 16384, 4096, 64
rng = np.random.default_rng(3)
for case in ("mixed-sign R", "positive R"):
    A = rng.integers(0, 32, (M, K)).astype(np.float32); S = rng.integers(0, 4, (K, N)).astype(np.float32)
    P = A.astype(np.float64) @ S.astype(np.float64)
    Y = (P + rng.integers(-400, 401, (M, N)) if case == "mixed-sign R" else P - rng.integers(1, 800, (M, N))).astype(np.float32)
    R = P - Y
    gA64, gS64 = R @ S.astype(np.float64).T, A.astype(np.float64).T @ R
    print(case, "max|gA| %.3g max|gS| %.3g (2^24 = 1.68e7)" % (np.abs(gA64).max(), np.abs(gS64).max()))
    for name, mode, env in (("f32", "f32", {}), ("f16x2", "f16x2", {}), ("f16x2r <HH>", "f16x2r", {}), ("f16x2r <R3>", "f16x2r", {"PMX_F16_R3": "1"}), ("f16x2 no chain", "f16x2", {"PMX_K1_CHAIN": "0"})):
        os.environ.update(env)
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A, S)
            gA, gS = dev.grad()
        for k in env: del os.environ[k]
        out = []
        for gq, r in ((gA, gA64), (gS, gS64)):
            e = gq.astype(np.float64) - r
            u = np.ldexp(1.0, np.floor(np.log2(np.maximum(np.abs(r), 1.0))).astype(int) - 23)
            out.append("mean %+.3f ulp, rms %.3f ulp, exact entries %.1f %%" % ((e / u).mean(), np.sqrt(((e / u) ** 2).mean()), 100.0 * (e == 0).mean()))
        print("  %-16s gA: %s | gS: %s" % (name, out[0], out[1]), flush=True)
