"""Randomized end-to-end sweep of nmf() against the fp64 oracle (helper of tests/test_gpu_fuzz.py and scratch/fuzz_nmf2.py -- not a test
module): shapes incl. ragged ones and any K, the three back-ends, the three arithmetic modes, weights, FISTA, the line search, every
operator of operators.py:20-160 with unity on either block, fp64 inputs of small problems.  Flags GROSS disagreement only -- a fraction
of entries out of tolerance that the reference's own fp32 arithmetic does not show, a NaN pattern that differs, an exception --:
adaprox's eps-clamp outliers are the yardstick's business (tests/test_gpu_parity_long.py).  Round 4 found two defects with it: mode
f16x2 on factors that ran away from the data (tests/test_gpu_range.py) and the step-rule dispatch of a weighted pgm."""
from functools import partial

import numpy as np


def pick_prox(rng, ops, block, allow_unity):
    """(library callable, oracle spec)"""
    k = int(rng.integers(0, 8 if allow_unity else 6))
    th = float(rng.choice([1e-3, 1e-2, 0.1]))
    if k == 0 or k == 1:
        return ops.prox_plus, ("plus",)
    if k == 2:
        return partial(ops.prox_soft_plus, thresh=th), ("soft_plus", th, "relative")
    if k == 3:
        return partial(ops.prox_soft, thresh=th), ("soft", th, "relative")
    if k == 4:
        return partial(ops.prox_hard_plus, thresh=th, type="absolute"), ("hard_plus", th, "absolute")
    if k == 5:
        return partial(ops.prox_max, thresh=5.0, type="absolute"), ("max", 5.0, "absolute")
    ax = 1 if block == 0 else 0          # rows of A / columns of S on the simplex
    return partial(ops.prox_unity_plus, axis=ax), ("unity_plus", ax)


def run(seed, n_cases, only=None, F64=False, log=print):
    """-> number of failing cases"""
    import proxmin_amd as pm
    from oracle import nmf_oracle as orc
    ops = pm.operators
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        kind = int(rng.integers(0, 6))
        if kind == 0:
            M, N, K = int(rng.integers(2, 600)), int(rng.integers(2, 600)), int(rng.integers(1, 33))
        elif kind == 1:
            M, N, K = 128 * int(rng.integers(1, 12)), 256 * int(rng.integers(1, 6)), int(rng.choice([32, 64, 128]))
        elif kind <= 3:
            M, N, K = int(rng.integers(300, 2200)), int(rng.integers(300, 2200)), int(rng.integers(2, 129))
        else:
            M, N, K = int(rng.integers(100, 1500)), int(rng.integers(100, 3000)), int(rng.choice([5, 16, 32, 64, 128]))
        if F64:
            M, N, K = int(rng.integers(2, 1000)), int(rng.integers(2, 1000)), int(rng.integers(1, 17))
        algo = ["pgm", "adaprox", "bsdmm"][int(rng.integers(0, 3))]
        mode = ["f32", "bf16x3", "f16x2"][int(rng.integers(0, 3))]
        its = int(rng.integers(2, 7))
        weighted = algo != "bsdmm" and rng.random() < 0.35
        unity_blk = int(rng.integers(0, 3))          # 2: none
        pA, sA = pick_prox(rng, ops, 0, unity_blk == 0)
        pS, sS = pick_prox(rng, ops, 1, unity_blk == 1)
        accel = algo == "pgm" and rng.random() < 0.4
        bt = algo == "pgm" and rng.random() < 0.3
        scheme = ["adam", "amsgrad", "nadam", "padam", "adamx"][int(rng.integers(0, 5))]
        sd = int(rng.integers(1 << 30))
        if only is not None and case not in only:
            continue
        DT = np.float64 if F64 else np.float32
        if F64:
            weighted = bt = False
            mode = "f32"                     # (the default mode: fp64 inputs of a small problem take the fp64 path by themselves)
        Y, A0, S0 = orc.synthetic_problem(M, N, K, DT, unity_S=(sS[0] == "unity_plus"), seed=sd)
        if sA[0] == "unity_plus":
            A0 = (A0 / A0.sum(axis=1, keepdims=True)).astype(DT)
        W = None
        if weighted:
            W = (0.1 + 2.0 * rng.random((M, N))).astype(np.float32)
            W[rng.random((M, N)) < 0.2] = 0
        desc = "%dx%dx%d %s %s its=%d W=%d accel=%d bt=%d proxA=%s proxS=%s%s" % (M, N, K, algo, mode, its, weighted, accel, bt, sA[0], sS[0], " " + scheme if algo == "adaprox" else "")
        pm.set_default_mode(mode)
        A, S = A0.copy(), S0.copy()
        Ao, So = A0.astype(np.float64), S0.astype(np.float64)
        Y64, W64 = Y.astype(np.float64), (None if W is None else W.astype(np.float64))
        kwW = {} if W is None else {"W": W}
        try:
            if algo == "pgm":
                sc = 0.5 if accel else 1.0
                step = pm.nmf.scaled_step_pgm(sc) if (accel or weighted) else None     # (the reference's default rule raises with a weight ARRAY: nmf.py:64)
                kwf = {"f": partial(pm.nmf.log_likelihood, Y=Y, **kwW)} if bt else {}      # (algorithms.py:59: the line search needs the smooth function)
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, accelerated=accel, backtracking=bt, step=step, max_iter=its, e_rel=1e-12, **kwW, **kwf)
                ostep = (lambda A_, S_, it=None, grads=None: tuple(sc * s for s in orc.lipschitz_steps(A_, S_))) if (accel or weighted) else None
                orc.pgm_nmf(Y64, Ao, So, sA, sS, step=ostep, accelerated=accel, backtracking=bt, max_iter=its, e_rel=1e-12, W=W64)
            elif algo == "bsdmm":
                g1, s1 = pick_prox(rng, ops, 0, False)
                g2, s2 = pick_prox(rng, ops, 1, False)
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.bsdmm, proxs_g=[[g1], [g2, ops.prox_plus]], max_iter=its, e_rel=1e-12)
                orc.bsdmm_nmf(Y64, Ao, So, sA, sS, proxs_g=[[s1], [s2, ("plus",)]], max_iter=its, e_rel=1e-12)
            else:
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.adaprox, scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False, **kwW)
                orc.adaprox_nmf(Y64, Ao, So, sA, sS, scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False, W=W64)
            ok = True; worst = 0.0; fr = 1.0
            for a, b in ((A, Ao), (S, So)):
                if not np.array_equal(np.isnan(a), np.isnan(b)):
                    ok = False
                fin = np.isfinite(b) & np.isfinite(a)
                a, b = a[fin], b[fin]
                if a.size == 0:
                    continue
                r = np.abs(a.astype(np.float64) - b) / ((1e-12 + 1e-9 * np.abs(b)) if F64 else (2e-5 + 2e-4 * np.abs(b)))
                worst = max(worst, float(r.max())); fr = min(fr, float((r <= 1).mean()))
            if (F64 and worst > 1) or fr < 0.99 or (algo != "adaprox" and worst > 50):
                ok = False
        except np.linalg.LinAlgError as e:          # the oracle's own eigen-solver on a NaN Gram matrix (the reference fails the same way)
            log("skip case %d %s: %s" % (case, desc, e))
            continue
        except Exception as e:
            ok = False; worst = float("nan"); fr = float("nan"); log("EXC", type(e).__name__, str(e)[:300])
        log("%s case %d %s: frac %.5f worst %.1f" % ("ok  " if ok else "FAIL", case, desc, fr, worst))
        bad += not ok
    pm.set_default_mode("f32")
    return bad

