# re-run chosen cases of fuzz_nmf.py (same RNG stream) and print the worst ratios for every library given in PMX_LIB
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
seed, n_cases, want = int(sys.argv[1]), int(sys.argv[2]), set(int(x) for x in sys.argv[3].split(","))
rng = np.random.default_rng(seed)
for case in range(n_cases):
    kind = rng.integers(0, 7)
    if kind == 0:
        M, N, K = int(rng.integers(2, 900)), int(rng.integers(2, 900)), int(rng.integers(1, 65))
    elif kind >= 5:
        M, N, K = int(rng.integers(500, 2500)), int(rng.integers(500, 2500)), int(rng.integers(17, 129))
    elif kind >= 3:
        M, N, K = int(rng.integers(500, 2500)), int(rng.integers(500, 2500)), int(rng.choice([32, 64, 128]))
    elif kind == 1:
        M, N, K = 128 * int(rng.integers(1, 10)), 256 * int(rng.integers(1, 6)), 64
    else:
        M, N, K = int(rng.integers(2, 1500)), 64 * int(rng.integers(1, 12)), int(rng.choice([32, 64, 100]))
    algo = ["pgm", "adaprox", "bsdmm"][int(rng.integers(0, 3))]
    unity = bool(rng.integers(0, 2)) and algo == "adaprox"
    mode = ["f32", "bf16x3", "f16x2"][int(rng.integers(0, 3))]
    its = int(rng.integers(2, 7))
    sd = int(rng.integers(1 << 30))
    scheme = None
    if algo == "adaprox":
        scheme = ["adam", "amsgrad", "nadam", "radam", "padam", "adamx"][int(rng.integers(0, 6))]
    if case not in want:
        continue
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=sd)
    pS = partial(pm.operators.prox_unity_plus, axis=0) if unity else pm.operators.prox_plus
    Ao, So = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0) if unity else ("plus",), scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False)
    A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
    orc.adaprox_nmf(Y.astype(np.float64), A64, S64, ("plus",), ("unity_plus", 0) if unity else ("plus",), scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False)
    for m, fr in (("f32", "1"), ("bf16x3", "1"), ("f16x2", "1"), ("f32", "0"), ("f16x2", "0")):
        os.environ["PMX_FRAME"] = fr
        pm.set_default_mode(m)
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme=scheme, prox_S=pS, max_iter=its, e_rel=1e-3, check_convergence=False)
        out = []
        for ref in ((Ao, So), (A64, S64)):
            worst, frac = 0.0, 1.0
            for a, b in ((A, ref[0]), (S, ref[1])):
                r = np.abs(a.astype(np.float64) - b) / (2e-5 + 2e-4 * np.abs(b))
                worst = max(worst, float(r.max())); frac = min(frac, float((r <= 1).mean()))
            out.append("worst %.1f frac %.5f" % (worst, frac))
        print("case %d %dx%dx%d %s unity=%d its=%d mode %s frame=%s: vs fp32 oracle %s | vs fp64 oracle %s" % (case, M, N, K, scheme, unity, its, m, fr, out[0], out[1]), flush=True)
    r = np.abs(Ao - A64) / (2e-5 + 2e-4 * np.abs(A64)); r2 = np.abs(So - S64) / (2e-5 + 2e-4 * np.abs(S64))
    print("   yardstick: fp32 oracle vs fp64 oracle worst %.1f" % max(r.max(), r2.max()))
