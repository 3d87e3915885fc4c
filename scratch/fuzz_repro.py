# replay scratch/fuzz_nmf.py's random stream up to one case and run that case in every mode / with switches (debugging aid)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from functools import partial
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    kind = rng.integers(0, 3)
    if kind == 0:
        M, N, K = int(rng.integers(2, 900)), int(rng.integers(2, 900)), int(rng.integers(1, 65))
    elif kind == 1:
        M, N, K = 128 * int(rng.integers(1, 10)), 256 * int(rng.integers(1, 6)), 64
    else:
        M, N, K = int(rng.integers(2, 1500)), 64 * int(rng.integers(1, 12)), int(rng.choice([32, 64, 100]))
    algo = ["pgm", "adaprox", "bsdmm"][int(rng.integers(0, 3))]
    unity = bool(rng.integers(0, 2)) and algo == "adaprox"
    mode = ["f32", "bf16x3", "f16x2"][int(rng.integers(0, 3))]
    its = int(rng.integers(2, 7))
    dseed = int(rng.integers(1 << 30))
    scheme = None
    if algo == "adaprox":
        scheme = ["adam", "amsgrad", "nadam", "radam", "padam", "adamx"][int(rng.integers(0, 6))]
print("case", target, M, N, K, algo, "unity", unity, mode, "its", its, "scheme", scheme, flush=True)
Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=dseed)
assert algo == "adaprox"
pS = partial(pm.operators.prox_unity_plus, axis=0) if unity else pm.operators.prox_plus
Ao, So = A0.copy(), S0.copy()
orc.adaprox_nmf(Y, Ao, So, ("plus",), ("unity_plus", 0) if unity else ("plus",), scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False)
print("oracle: |A| max %.3e  |S| max %.3e  finite %s" % (np.abs(Ao).max(), np.abs(So).max(), np.isfinite(Ao).all() and np.isfinite(So).all()))
for env in ({}, {"PMX_TAIL_FUSED": "0"}, {"PMX_K1_CHAIN": "0"}):
    for k, v in env.items():
        os.environ[k] = v
    for md in ("f32", "bf16x3", "f16x2"):
        pm.set_default_mode(md)
        for n in range(1, its + 1):
            A, S = A0.copy(), S0.copy()
            pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, scheme=scheme, prox_S=pS, max_iter=n, e_rel=1e-3, check_convergence=False)
            Ar, Sr = A0.copy(), S0.copy()
            orc.adaprox_nmf(Y, Ar, Sr, ("plus",), ("unity_plus", 0) if unity else ("plus",), scheme=scheme, max_iter=n, e_rel=1e-3, check_convergence=False)
            rA = (np.abs(A - Ar) / (2e-5 + 2e-4 * np.abs(Ar))).max(); rS = (np.abs(S - Sr) / (2e-5 + 2e-4 * np.abs(Sr))).max()
            print("  env %s mode %s its %d: worst ratio A %.3g S %.3g   max|A| %.3e max|S| %.3e" % (env, md, n, rA, rS, np.abs(A).max(), np.abs(S).max()), flush=True)
    for k in env:
        del os.environ[k]
