# scratch: randomized end-to-end sweep: nmf() in all three back-ends / arithmetic modes vs the oracle, a few iterations
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from functools import partial
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(n_cases):
    kind = rng.integers(0, 7)
    if kind == 0:
        M, N, K = int(rng.integers(2, 900)), int(rng.integers(2, 900)), int(rng.integers(1, 65))
    elif kind >= 5:    # [r4] any K on a shape large enough for the matrix-core kernels: the frame pads the components too
        M, N, K = int(rng.integers(500, 2500)), int(rng.integers(500, 2500)), int(rng.integers(17, 129))
    elif kind >= 3:    # [r4] ragged shapes with a tuned K: the zero-padded frame
        M, N, K = int(rng.integers(500, 2500)), int(rng.integers(500, 2500)), int(rng.choice([32, 64, 128]))
    elif kind == 1:
        M, N, K = 128 * int(rng.integers(1, 10)), 256 * int(rng.integers(1, 6)), 64
    else:
        M, N, K = int(rng.integers(2, 1500)), 64 * int(rng.integers(1, 12)), int(rng.choice([32, 64, 100]))
    algo = ["pgm", "adaprox", "bsdmm"][int(rng.integers(0, 3))]
    unity = bool(rng.integers(0, 2)) and algo == "adaprox"
    mode = ["f32", "bf16x3", "f16x2"][int(rng.integers(0, 3))]
    its = int(rng.integers(2, 7))
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=int(rng.integers(1 << 30)))
    pm.set_default_mode(mode)
    A, S = A0.copy(), S0.copy(); Ao, So = A0.copy(), S0.copy()
    try:
        if algo == "pgm":
            pm.nmf.nmf(Y, A, S, max_iter=its, e_rel=1e-12)
            orc.pgm_nmf(Y, Ao, So, max_iter=its, e_rel=1e-12)
        elif algo == "bsdmm":
            pg = [[pm.operators.prox_plus, partial(pm.operators.prox_soft, thresh=1e-3)]] * 2
            pm.nmf.nmf(Y, A, S, algorithm=pm.bsdmm, proxs_g=pg, max_iter=its, e_rel=1e-12)
            orc.bsdmm_nmf(Y, Ao, So, proxs_g=[[("plus",), ("soft", 1e-3, "relative")]] * 2, max_iter=its, e_rel=1e-12)
        else:
            scheme = ["adam", "amsgrad", "nadam", "radam", "padam", "adamx"][int(rng.integers(0, 6))]
            pS = partial(pm.operators.prox_unity_plus, axis=0) if unity else pm.operators.prox_plus
            pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme=scheme, prox_S=pS, max_iter=its, e_rel=1e-3, check_convergence=False)
            orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0) if unity else ("plus",), scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False)
        ok = True; worst = 0.0
        for a, b in ((A, Ao), (S, So)):
            if not np.array_equal(np.isnan(a), np.isnan(b)):      # the reference's own NaNs (e.g. prox_unity on an all-zero column) must match
                ok = False
            fin = np.isfinite(b)
            a, b = a[fin], b[fin]
            if a.size == 0:
                continue
            r = np.abs(a.astype(np.float64) - b) / (2e-5 + 2e-4 * np.abs(b))
            worst = max(worst, float(r.max()))
            if (r <= 1).mean() < 0.995 or r.max() > 50:
                ok = False
    except Exception as e:
        ok = False; worst = float("nan"); print("EXC", repr(e))
    if not ok:
        bad += 1
        print("FAIL case %d %dx%dx%d %s unity=%d %s its=%d worst ratio %.1f" % (case, M, N, K, algo, unity, mode, its, worst), flush=True)
print("fuzz done: %d cases, %d failures" % (n_cases, bad), flush=True)
