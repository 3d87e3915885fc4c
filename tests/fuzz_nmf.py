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
        case_seed = int(rng.integers(1 << 62))        # every case draws from its own generator: replaying a subset (`only`) gives the same cases
        if only is not None and case not in only:
            continue
        crng = np.random.default_rng(case_seed)
        kind = int(crng.integers(0, 6))
        if kind == 0:
            M, N, K = int(crng.integers(2, 600)), int(crng.integers(2, 600)), int(crng.integers(1, 33))
        elif kind == 1:
            M, N, K = 128 * int(crng.integers(1, 12)), 256 * int(crng.integers(1, 6)), int(crng.choice([32, 64, 128]))
        elif kind <= 3:
            M, N, K = int(crng.integers(300, 2200)), int(crng.integers(300, 2200)), int(crng.integers(2, 129))
        else:
            M, N, K = int(crng.integers(100, 1500)), int(crng.integers(100, 3000)), int(crng.choice([5, 16, 32, 64, 128]))
        if F64 == "big":                     # [r6] fp64 at size (k_big_f64.hip): the shapes drawn above, K up to 128
            pass
        elif F64:
            M, N, K = int(crng.integers(2, 1000)), int(crng.integers(2, 1000)), int(crng.integers(1, 17))
        algo = ["pgm", "adaprox", "bsdmm"][int(crng.integers(0, 3))]
        mode = ["f32", "bf16x3", "f16x2", "f16x2r"][int(crng.integers(0, 4))]
        its = int(crng.integers(2, 7))
        weighted = algo != "bsdmm" and crng.random() < 0.35
        unity_blk = int(crng.integers(0, 3))          # 2: none
        pA, sA = pick_prox(crng, ops, 0, unity_blk == 0)
        pS, sS = pick_prox(crng, ops, 1, unity_blk == 1)
        accel = algo == "pgm" and crng.random() < 0.4
        bt = algo == "pgm" and crng.random() < 0.3
        scheme = ["adam", "amsgrad", "nadam", "padam", "adamx"][int(crng.integers(0, 5))]
        sd = int(crng.integers(1 << 30))
        DT = np.float64 if F64 else np.float32
        if F64:
            weighted = weighted and F64 == "big"       # (the matrix-core kernels take weights)
            bt = False
            mode = "f32"                     # (the default mode: fp64 inputs of a small problem take the fp64 path by themselves)
        Y, A0, S0 = orc.synthetic_problem(M, N, K, DT, unity_S=(sS[0] == "unity_plus"), seed=sd)
        if sA[0] == "unity_plus":
            A0 = (A0 / A0.sum(axis=1, keepdims=True)).astype(DT)
        W = None
        if weighted:
            W = (0.1 + 2.0 * crng.random((M, N))).astype(DT)
            W[crng.random((M, N)) < 0.2] = 0
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
                g1, s1 = pick_prox(crng, ops, 0, False)
                g2, s2 = pick_prox(crng, ops, 1, False)
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.bsdmm, proxs_g=[[g1], [g2, ops.prox_plus]], max_iter=its, e_rel=1e-12)
                orc.bsdmm_nmf(Y64, Ao, So, sA, sS, proxs_g=[[s1], [s2, ("plus",)]], max_iter=its, e_rel=1e-12)
            else:
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.adaprox, scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False, **kwW)
                orc.adaprox_nmf(Y64, Ao, So, sA, sS, scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False, W=W64)
            ok = True; worst = 0.0; fr = 1.0
            nan_differs = False
            for a, b in ((A, Ao), (S, So)):
                if not np.array_equal(np.isnan(a), np.isnan(b)):
                    nan_differs = True
                    desc += " [NaN pattern: device %d, oracle %d of %d]" % (int(np.isnan(a).sum()), int(np.isnan(b).sum()), a.size)
            if nan_differs and algo == "adaprox" and not F64:
                # a run that DIVERGES (cold-start PAdam: Psi = V^p, steps of 1e3) overflows fp32 where fp64 still counts: the reference computes
                # fp32 inputs in fp32 (nmf.py:39-41), so the yardstick for the NaN pattern is the oracle run in fp32
                A32, S32 = A0.copy(), S0.copy()
                orc.adaprox_nmf(Y, A32, S32, sA, sS, scheme=scheme, max_iter=its, e_rel=1e-3, check_convergence=False, W=W)
                same = np.array_equal(np.isnan(A), np.isnan(A32)) and np.array_equal(np.isnan(S), np.isnan(S32))
                desc += " [fp32 oracle: NaN %d + %d -> %s]" % (int(np.isnan(A32).sum()), int(np.isnan(S32).sum()), "same as the device" if same else "differs")
                nan_differs = not same
            if nan_differs:
                ok = False
            for a, b in ((A, Ao), (S, So)):
                fin = np.isfinite(b) & np.isfinite(a)
                a, b = a[fin], b[fin]
                if a.size == 0:
                    continue
                r = np.abs(a.astype(np.float64) - b) / ((1e-12 + 1e-9 * np.abs(b) + (1e-10 * float(np.abs(b).max()) if F64 == "big" else 0.0)) if F64 else (2e-5 + 2e-4 * np.abs(b)))
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
    pm.set_default_mode(None)
    return bad



def run_options(seed, n_cases, only=None, log=print):
    """Second sweep: the ARGUMENTS of the three back-ends rather than shapes and operators -- stopping tests that fire (loose e_rel,
    bsdmm's e_abs: the iteration the run ends at must be the oracle's), adaprox's b1 as an array, b2 / eps / p, a capped proximal loop
    (prox_max_iter 1 .. 3), warm-started moments, prox=None on a block (algorithms.py:380), bsdmm with constraints on one block only.
    -> number of failing cases"""
    import proxmin_amd as pm
    from oracle import nmf_oracle as orc
    ops = pm.operators
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        case_seed = int(rng.integers(1 << 62))        # every case draws from its own generator: replaying a subset (`only`) gives the same cases
        if only is not None and case not in only:
            continue
        crng = np.random.default_rng(case_seed)
        big = crng.random() < 0.4
        if big:
            M, N, K = int(crng.integers(300, 1800)), int(crng.integers(300, 1800)), int(crng.choice([8, 32, 50, 64, 128]))
        else:
            M, N, K = int(crng.integers(5, 500)), int(crng.integers(5, 500)), int(crng.integers(1, 17))
        algo = ["pgm", "adaprox", "bsdmm"][int(crng.integers(0, 3))]
        mode = ["f32", "f16x2", "f16x2r"][int(crng.integers(0, 3))]
        e_rel = float(crng.choice([1e-2, 3e-2, 1e-3]))
        max_iter = int(crng.integers(5, 40))
        unity = crng.random() < 0.3
        scheme = ["adam", "amsgrad", "nadam", "padam", "adamx"][int(crng.integers(0, 5))]
        b1_kind, b2, eps, pp = int(crng.integers(0, 3)), float(crng.choice([0.999, 0.99])), float(crng.choice([1e-8, 1e-6])), float(crng.choice([0.25, 0.125]))
        pmi = int(crng.choice([1000, 1000, 3, 1]))
        check = crng.random() < 0.7
        none_blk = int(crng.integers(0, 4))           # 0 / 1: prox=None on that block (adaprox), >= 2: none
        accel = crng.random() < 0.4
        e_abs = float(crng.choice([0.0, 0.0, 1e-4]))
        g_kind = int(crng.integers(0, 3))
        sd = int(crng.integers(1 << 30))
        Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=sd)
        pS, sS = (partial(ops.prox_unity_plus, axis=0), ("unity_plus", 0)) if unity else (ops.prox_plus, ("plus",))
        pA, sA = ops.prox_plus, ("plus",)
        desc = "%dx%dx%d %s %s e_rel=%g max_iter=%d unity=%d" % (M, N, K, algo, mode, e_rel, max_iter, unity)
        counts = {"dev": 0, "orc": 0}

        def counter(key):
            def cb(*X, it=None):
                counts[key] += 1
            return cb
        pm.set_default_mode(mode)
        A, S = A0.copy(), S0.copy()
        Ao, So = A0.astype(np.float64), S0.astype(np.float64)
        Y64 = Y.astype(np.float64)
        try:
            if algo == "pgm":
                c = 0.5 if accel else 1.0
                desc += " accel=%d" % accel
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, accelerated=accel, step=pm.nmf.scaled_step_pgm(c), max_iter=max_iter, e_rel=e_rel, callback=counter("dev"))
                orc.pgm_nmf(Y64, Ao, So, sA, sS, step=lambda A_, S_, it=None, grads=None: tuple(c * s for s in orc.lipschitz_steps(A_, S_)),
                            accelerated=accel, max_iter=max_iter, e_rel=e_rel, callback=counter("orc"))
            elif algo == "adaprox":
                b1 = [0.9, 0.8, 0.9 * 0.98 ** np.arange(max_iter)][b1_kind]
                if none_blk == 0:
                    pA, sA = None, None
                elif none_blk == 1 and not unity:
                    pS, sS = None, None
                warm = crng.random() < 0.3
                kw = dict(scheme=scheme, b1=b1, b2=b2, eps=eps, p=pp, prox_max_iter=pmi, check_convergence=check, max_iter=max_iter, e_rel=e_rel)
                desc += " %s b1kind=%d b2=%g eps=%g p=%g pmi=%d check=%d none=%d warm=%d" % (scheme, b1_kind, b2, eps, pp, pmi, check, none_blk, warm)
                kwd, kwo, kw32 = {}, {}, {}
                if warm:
                    m0 = [np.full(A0.shape, 0.01, np.float32), np.full(S0.shape, -0.02, np.float32)]
                    v0 = [np.full(A0.shape, 0.5, np.float32), np.full(S0.shape, 0.25, np.float32)]
                    kwd = dict(M=[x.copy() for x in m0], V=[x.copy() for x in v0])
                    kwo = dict(M=[x.astype(np.float64) for x in m0], V=[x.astype(np.float64) for x in v0])
                    kw32 = dict(M=[x.copy() for x in m0], V=[x.copy() for x in v0])
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.adaprox, callback=counter("dev"), **kw, **kwd)
                orc.adaprox_nmf(Y64, Ao, So, sA, sS, callback=counter("orc"), **kw, **kwo)
            else:
                soft = (partial(ops.prox_soft, thresh=1e-3), ("soft", 1e-3, "relative"))
                gl = [[[ops.prox_plus], [soft[0], ops.prox_plus]], [None, [ops.prox_plus]], [[soft[0]], None]][g_kind]
                go = [[[("plus",)], [soft[1], ("plus",)]], [None, [("plus",)]], [[soft[1]], None]][g_kind]
                desc += " e_abs=%g g=%d" % (e_abs, g_kind)
                pm.nmf.nmf(Y, A, S, prox_A=pA, prox_S=pS, algorithm=pm.bsdmm, proxs_g=gl, max_iter=max_iter, e_rel=e_rel, e_abs=e_abs, callback=counter("dev"))
                orc.bsdmm_nmf(Y64, Ao, So, sA, sS, proxs_g=go, max_iter=max_iter, e_rel=e_rel, e_abs=e_abs, callback=counter("orc"))
            ok = True
            worst, fr = 0.0, 1.0
            dn = counts["dev"] - counts["orc"]
            if abs(dn) > 1:
                # the stopping test fired at another iteration.  Yardstick: the oracle itself in fp32 -- a test that compares two sums
                # that are EXACTLY equal in fp64 (K = 1 with unity columns: S == 1 for ever, Boyd's dual residual 0 <= 0) fires in fp64
                # and not in any fp32 arithmetic: the case counts only when the fp32 oracle ends where the fp64 one does
                counts["o32"] = 0
                A32, S32 = A0.copy(), S0.copy()
                if algo == "pgm":
                    orc.pgm_nmf(Y, A32, S32, sA, sS, step=lambda A_, S_, it=None, grads=None: tuple(c * s for s in orc.lipschitz_steps(A_, S_)),
                                accelerated=accel, max_iter=max_iter, e_rel=e_rel, callback=counter("o32"))
                elif algo == "adaprox":
                    orc.adaprox_nmf(Y, A32, S32, sA, sS, callback=counter("o32"), **kw, **kw32)
                else:
                    orc.bsdmm_nmf(Y, A32, S32, sA, sS, proxs_g=go, max_iter=max_iter, e_rel=e_rel, e_abs=e_abs, callback=counter("o32"))
                desc += " [fp32 oracle: %d callbacks]" % counts["o32"]
                ok = abs(counts["o32"] - counts["orc"]) > 1 and abs(counts["dev"] - counts["o32"]) <= 1
            elif dn == 0:
                for a, b in ((A, Ao), (S, So)):
                    if not np.array_equal(np.isnan(a), np.isnan(b)):
                        ok = False
                    fin = np.isfinite(b) & np.isfinite(a)
                    a, b = a[fin], b[fin]
                    if a.size == 0:
                        continue
                    r = np.abs(a.astype(np.float64) - b) / (2e-5 + 2e-4 * np.abs(b))
                    worst = max(worst, float(r.max())); fr = min(fr, float((r <= 1).mean()))
                if fr < 0.99 or (algo != "adaprox" and worst > 50):
                    ok = False
                if not ok and algo == "adaprox" and fr == fr:
                    # a long adaprox run can be chaotic in ANY fp32 arithmetic (entries whose second moment is at eps): the oracle's own
                    # fp32 run against its fp64 run is the yardstick -- the case only counts when the device is clearly worse than that
                    A32, S32 = A0.copy(), S0.copy()
                    orc.adaprox_nmf(Y, A32, S32, sA, sS, **kw, **kw32)
                    fy = min(float((np.abs(a_.astype(np.float64) - b_) <= 2e-5 + 2e-4 * np.abs(b_)).mean()) for a_, b_ in ((A32, Ao), (S32, So)))
                    wy = max(float((np.abs(a_.astype(np.float64) - b_) / (2e-5 + 2e-4 * np.abs(b_))).max()) for a_, b_ in ((A32, Ao), (S32, So)))
                    desc += " [yardstick frac %.5f worst %.1f]" % (fy, wy)
                    # (either the fractions are comparable, or the whole error is: a run that amplifies ANY fp32 noise a thousandfold -- AMSGrad with a
                    # decaying b1 over 36 iterations: scratch/r4_case133.py -- leaves the fp32 oracle at 0.8 x the bound everywhere and the device at 3 x)
                    if 1.0 - fr <= 3.0 * (1.0 - fy) + 1e-3 or worst <= 6.0 * max(wy, 0.5):
                        ok = True
        except np.linalg.LinAlgError as e:
            log("skip case %d %s: %s" % (case, desc, e))
            continue
        except Exception as e:
            ok = False; worst = fr = float("nan"); dn = 0
            log("EXC %s %s" % (type(e).__name__, str(e)[:300]))
        log("%s case %d %s: callbacks %d / %d frac %.5f worst %.1f" % ("ok  " if ok else "FAIL", case, desc, counts["dev"], counts["orc"], fr, worst))
        bad += not ok
    pm.set_default_mode(None)
    return bad
