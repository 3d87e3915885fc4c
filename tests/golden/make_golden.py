#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (it needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

It imports `proxmin` v0.6.12 from /root/reference, runs seeded problems through
`proxmin.nmf.nmf()` and the helper functions on the nmf() path, and writes inputs + outputs as
`.npz` files next to this script.  The fixtures are DATA (inputs and expected outputs); no
reference source text is stored.  Constraints are recorded as prox-spec tuples encoded in JSON
(see oracle/nmf_oracle.py) so that neither the oracle nor the product has to import the
reference to read them.
"""
import importlib.util
import json
import os
import sys
from functools import partial

import numpy as np

REF = os.environ.get("PROXMIN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import proxmin  # noqa: E402
from proxmin import algorithms as ralg, nmf as rnmf, operators as rops, utils as rutils  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def spec_to_callable(spec):
    """prox-spec tuple -> reference callable."""
    if spec is None:
        return None
    name = spec[0]
    fn = getattr(rops, "prox_" + name)
    if name in ("unity", "unity_plus"):
        return partial(fn, axis=spec[1])
    if name in ("min", "max", "hard", "hard_plus", "soft", "soft_plus"):
        kw = {"thresh": spec[1]}
        if len(spec) > 2:
            kw["type"] = spec[2]
        return partial(fn, **kw)
    return fn


def problem(M, N, K, dtype, unity_S, seed):
    """Same recipe as oracle.nmf_oracle.synthetic_problem (kept independent on purpose)."""
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        At, St = rng.random((M, K), dtype=np.float32), rng.random((K, N), dtype=np.float32)
    else:
        At, St = rng.random((M, K)), rng.random((K, N))
    if unity_S:
        St /= St.sum(0, keepdims=True)
    Y = (At @ St + 0.01 * rng.standard_normal((M, N))).astype(dtype)
    A0 = rng.random((M, K)).astype(dtype)
    S0 = rng.random((K, N)).astype(dtype)
    if unity_S:
        S0 /= S0.sum(0, keepdims=True)
    return Y, A0, S0


def half_step(*X, it=None):
    return tuple(0.5 * s for s in rnmf.step_pgm(*X))


# every case: name -> dict(algorithm, prox_A, prox_S, kwargs(json-able), unity_S)
CASES = {
    "pgm": dict(alg="pgm", pA=("plus",), pS=("plus",), kw={}),
    "fista_half": dict(alg="pgm", pA=("plus",), pS=("plus",), kw={"accelerated": True}, half=True),
    "pgm_unityA": dict(alg="pgm", pA=("unity_plus", 1), pS=("plus",), kw={}),
    "pgm_softS": dict(alg="pgm", pA=("plus",), pS=("soft_plus", 0.05, "relative"), kw={}),
    "adam": dict(alg="adaprox", pA=("plus",), pS=("plus",), kw={"scheme": "adam"}),
    "nadam": dict(alg="adaprox", pA=("plus",), pS=("plus",), kw={"scheme": "nadam"}),
    "amsgrad": dict(alg="adaprox", pA=("plus",), pS=("plus",), kw={"scheme": "amsgrad"}),
    "padam": dict(alg="adaprox", pA=("plus",), pS=("plus",), kw={"scheme": "padam"}),
    "adamx": dict(alg="adaprox", pA=("plus",), pS=("plus",), kw={"scheme": "adamx"}),
    "radam": dict(alg="adaprox", pA=("plus",), pS=("plus",), kw={"scheme": "radam"}),
    "amsgrad_unityS": dict(alg="adaprox", pA=("plus",), pS=("unity_plus", 0), kw={"scheme": "amsgrad"}, unity_S=True),
    "adam_unityA_softS": dict(alg="adaprox", pA=("unity_plus", 1), pS=("soft_plus", 0.1, "relative"), kw={"scheme": "adam"}),
    "amsgrad_noprox": dict(alg="adaprox", pA=None, pS=None, kw={"scheme": "amsgrad"}),
    "bsdmm_none": dict(alg="bsdmm", pA=("plus",), pS=("plus",), kw={}),
    "bsdmm_plus_soft": dict(alg="bsdmm", pA=("plus",), pS=("plus",), kw={},
                            proxs_g=[[("plus",), ("soft", 0.01, "relative")]] * 2),
    "bsdmm_mixed": dict(alg="bsdmm", pA=("plus",), pS=("plus",), kw={},
                        proxs_g=[None, [("soft_plus", 0.02, "relative")]]),
}


def run_case(case, Y, A0, S0, max_iter, e_rel, n_trace=6):
    alg = {"pgm": ralg.pgm, "adaprox": ralg.adaprox, "bsdmm": ralg.bsdmm}[case["alg"]]
    A, S = A0.copy(), S0.copy()
    tb = rutils.Traceback()
    kw = dict(case["kw"])
    if case.get("half"):
        kw["step"] = half_step
    if case.get("proxs_g") is not None:
        kw["proxs_g"] = [None if g is None else [spec_to_callable(s) for s in g] for g in case["proxs_g"]]
    if case["alg"] == "adaprox":
        kw.setdefault("check_convergence", True)
    ret = rnmf.nmf(Y, A, S, prox_A=spec_to_callable(case["pA"]), prox_S=spec_to_callable(case["pS"]),
                   algorithm=alg, max_iter=max_iter, e_rel=e_rel, callback=tb, **kw)
    out = {"A": A, "S": S, "loss": rnmf.log_likelihood(A, S, Y=Y), "n_callbacks": len(tb.trace)}
    # iterates seen by the callback (pre-update state) for the first few iterations
    for i, (At, St) in enumerate(tb.trace[:n_trace]):
        out["trace_A_%d" % i] = At
        out["trace_S_%d" % i] = St
    if case["alg"] == "pgm":
        conv, G, steps = ret
        out.update(conv=np.array(conv, dtype=bool), G_A=G[0], G_S=G[1], steps=np.array(steps, dtype=np.float64))
    elif case["alg"] == "adaprox":
        conv, Mm, Vv, Vh = ret
        out.update(conv=np.array([bool(c) if c is not None else False for c in conv]),
                   M_A=Mm[0], M_S=Mm[1], V_A=Vv[0], V_S=Vv[1],
                   vhat_none=np.array([v is None for v in Vh]))
    else:
        out.update(conv=np.array([bool(c) for c in ret]))
    return out


def write_nmf_fixture(fname, M, N, K, dtype, max_iter, e_rel, store_inputs, names, n_trace=6):
    blob = {}
    meta = {"M": M, "N": N, "K": K, "dtype": np.dtype(dtype).name, "max_iter": max_iter, "e_rel": e_rel,
            "seed": 1234, "cases": {}, "numpy": np.__version__, "reference": "proxmin 0.6.12"}
    probs = {}
    for name in names:
        case = CASES[name]
        u = bool(case.get("unity_S"))
        if u not in probs:
            probs[u] = problem(M, N, K, dtype, u, 1234)
        Y, A0, S0 = probs[u]
        out = run_case(case, Y, A0, S0, max_iter, e_rel, n_trace)
        for k, v in out.items():
            blob["%s/%s" % (name, k)] = np.asarray(v)
        meta["cases"][name] = {"alg": case["alg"], "prox_A": case["pA"], "prox_S": case["pS"], "kw": case["kw"],
                               "half_step": bool(case.get("half")), "proxs_g": case.get("proxs_g"), "unity_S": u}
        print("  %-22s loss=%.9g sumA=%.10g sumS=%.10g its=%d" % (name, out["loss"], out["A"].sum(), out["S"].sum(), out["n_callbacks"]))
    for u, (Y, A0, S0) in probs.items():
        tag = "unity" if u else "plain"
        if store_inputs:
            blob["inputs_%s/Y" % tag], blob["inputs_%s/A0" % tag], blob["inputs_%s/S0" % tag] = Y, A0, S0
        # checksums so a test that REGENERATES the inputs from the seed can detect generator drift
        blob["inputs_%s/checksum" % tag] = np.array([Y.sum(dtype=np.float64), A0.sum(dtype=np.float64), S0.sum(dtype=np.float64),
                                                     float(Y[0, 0]), float(Y[-1, -1])])
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, fname), **blob)
    print("wrote", fname)


def write_operator_fixture():
    rng = np.random.default_rng(77)
    blob, meta = {}, {"entries": []}
    shapes = {"A": (12, 5), "S": (5, 17)}
    steps = {"scalar": 0.37, "vecA": rng.random(5) * 0.5, "vecS": rng.random((5, 1)) * 0.5}
    specs = [("id",), ("zero",), ("plus",), ("unity", 0), ("unity", 1), ("unity_plus", 0), ("unity_plus", 1),
             ("min", 0.3, "relative"), ("min", -0.2, "absolute"), ("max", 0.3, "relative"), ("max", 0.1, "absolute"),
             ("hard", 0.8, "relative"), ("hard", 0.25, "absolute"), ("hard_plus", 0.8, "relative"),
             ("soft", 0.6, "relative"), ("soft", 0.2, "absolute"), ("soft_plus", 0.6, "relative"), ("soft_plus", 0.2, "absolute")]
    idx = 0
    for blk, shp in shapes.items():
        for dt in (np.float64, np.float32):
            X = (rng.standard_normal(shp) * 0.7 + 0.2).astype(dt)
            for sname in ("scalar", "vec" + blk):
                st = steps[sname]
                for spec in specs:
                    if sname != "scalar" and spec[0] in ("min", "max") :
                        continue  # the reference's masked assignment cannot broadcast an array threshold (raises)
                    out = spec_to_callable(spec)(X.copy(), st)
                    key = "op%03d" % idx
                    blob[key + "/X"], blob[key + "/out"] = X, out
                    blob[key + "/step"] = np.asarray(st, dtype=np.float64)
                    meta["entries"].append({"key": key, "spec": spec, "block": blk, "step": sname})
                    idx += 1
    # AlternatingProjections: list applied last-to-first, repeat times
    X = rng.standard_normal((9, 4))
    ap = rops.AlternatingProjections([partial(rops.prox_unity, axis=1), rops.prox_plus], repeat=3)
    blob["ap/X"], blob["ap/out"] = X, ap(X.copy(), 0.5)
    meta["ap"] = {"specs": [("unity", 1), ("plus",)], "repeat": 3, "step": 0.5}
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **blob)
    print("wrote operators.npz (%d entries)" % idx)


def write_helper_fixture():
    rng = np.random.default_rng(99)
    blob, meta = {}, {}
    # Nesterov omega sequence
    acc = rutils.NesterovAccelerator(accelerated=True)
    blob["nesterov/omega"] = np.array([acc.omega for _ in range(40)])
    # moment schemes at it in {0,1,5}, Vhat None and array, fp64 + fp32
    fns = {"adam": ralg._adam_phi_psi, "nadam": ralg._nadam_phi_psi, "amsgrad": ralg._amsgrad_phi_psi,
           "padam": ralg._padam_phi_psi, "adamx": ralg._adamx_phi_psi, "radam": ralg._radam_phi_psi}
    ents = []
    n = 0
    b1 = np.array([0.9 - 0.01 * i for i in range(10)])
    for sch, fn in fns.items():
        for it in (0, 1, 5):
            for with_vhat in (False, True):
                for dt in (np.float64, np.float32):
                    G = rng.standard_normal((6, 4)).astype(dt)
                    M0 = (rng.standard_normal((6, 4)) * 0.1).astype(dt)
                    V0 = (rng.random((6, 4)) * 0.01).astype(dt)
                    Vh0 = (rng.random((6, 4)) * 0.02).astype(dt) if with_vhat else None
                    M, V = M0.copy(), V0.copy()
                    Vh = None if Vh0 is None else Vh0.copy()
                    Phi, Psi = fn(it, G, M, V, Vh, b1, 0.999, 1e-8, 0.25)
                    key = "mom%03d" % n
                    n += 1
                    for nm, arr in (("G", G), ("M0", M0), ("V0", V0), ("M1", M), ("V1", V), ("Phi", np.asarray(Phi)), ("Psi", np.asarray(Psi))):
                        blob["%s/%s" % (key, nm)] = arr
                    if with_vhat:
                        blob[key + "/Vh0"], blob[key + "/Vh1"] = Vh0, Vh
                    ents.append({"key": key, "scheme": sch, "it": it, "vhat": with_vhat})
    blob["moments/b1"] = b1
    meta["moments"] = ents
    # spectral norm / step rules
    A = rng.random((40, 6))
    S = rng.random((6, 70))
    blob["steps/A"], blob["steps/S"] = A, S
    blob["steps/pgm"] = np.array(rnmf.step_pgm(A, S), dtype=np.float64)
    aA, aS = rnmf.step_adaprox(A, S)
    blob["steps/ada_A"], blob["steps/ada_S"] = aA, aS
    gA, gS = rnmf.grad_likelihood(A, S, Y=(A @ S + 0.1 * rng.standard_normal((40, 70))))
    # Barzilai-Borwein stepper, both types, standalone, 6 calls on a drifting sequence
    for typ in (1, 2):
        bb = rutils.BarzilaiBorweinStepper(type=typ, init_r=0.1)
        Xa, Xs = rng.random((8, 3)), rng.random((3, 9))
        outs = []
        for it in range(6):
            Ga, Gs = rng.standard_normal((8, 3)) * (1 + it), rng.standard_normal((3, 9))
            blob["bb%d/X_A_%d" % (typ, it)], blob["bb%d/X_S_%d" % (typ, it)] = Xa.copy(), Xs.copy()
            blob["bb%d/G_A_%d" % (typ, it)], blob["bb%d/G_S_%d" % (typ, it)] = Ga, Gs
            outs.append(np.asarray(bb.step(Xa, Xs, it=it, grads=(Ga, Gs)), dtype=np.float64))
            Xa = Xa - 0.05 * Ga
            Xs = Xs - 0.05 * Gs
        blob["bb%d/steps" % typ] = np.array(outs)
    # update_variables / do_the_mm / errors with identity adapters, two constraints
    X = rng.standard_normal((7, 5))
    L = [rutils.MatrixAdapter(None), rutils.MatrixAdapter(None)]
    Z, U = rutils.initZU(X, L)
    U[0][:] = rng.standard_normal(X.shape) * 0.1
    Z[1][:] = Z[1] + rng.standard_normal(X.shape) * 0.05
    G = rng.standard_normal(X.shape)
    sf = 0.3
    sg = [rutils.get_step_g(sf, 1, N=2, M=2)] * 2
    blob["uv/X0"], blob["uv/G"] = X.copy(), G
    blob["uv/Z0_0"], blob["uv/Z0_1"], blob["uv/U0_0"], blob["uv/U0_1"] = Z[0].copy(), Z[1].copy(), U[0].copy(), U[1].copy()
    prox_f = lambda Xx, step: rops.prox_plus(Xx - step * G, step)  # noqa: E731
    pg = [rops.prox_plus, partial(rops.prox_soft, thresh=0.4)]
    LX, R, Sd = rutils.update_variables(X, Z, U, prox_f, sf, pg, sg, L)
    conv, errs = rutils.check_constraint_convergence(X, L, LX, Z, U, R, Sd, sf, sg, 1e-2, 1e-3)
    blob["uv/X1"] = X
    for i in range(2):
        blob["uv/Z1_%d" % i], blob["uv/U1_%d" % i], blob["uv/R_%d" % i], blob["uv/Sd_%d" % i] = Z[i], U[i], R[i], Sd[i]
    blob["uv/errors"] = np.array(errs, dtype=np.float64)
    meta["uv"] = {"step_f": sf, "step_g": sg, "prox_f": ("plus",), "proxs_g": [("plus",), ("soft", 0.4, "relative")],
                  "e_rel": 1e-2, "e_abs": 1e-3, "converged": bool(conv)}
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **blob)
    print("wrote helpers.npz")


def write_unmixing_fixture():
    """examples/unmixing.py known answers: its own data generator (imported, not copied) with
    np.random.seed(101), then the reference solvers through nmf()-equivalent direct calls."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    spec = importlib.util.spec_from_file_location("ref_unmixing", os.path.join(REF, "examples", "unmixing.py"))
    um = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(um)
    n, k, b, noise = 50, 3, 100, 0.02
    np.random.seed(101)
    trueA = np.array([um.generateAmplitudes(k) for _ in range(b)])
    trueS = np.array([um.generateComponent(n) for _ in range(k)])
    Y = um.add_noise(np.dot(trueA, trueS), noise)
    A0 = np.random.rand(b, k)
    A0 /= A0.sum(axis=1)[:, None]
    S0 = np.random.rand(k, n)
    blob = {"Y": Y, "A0": A0, "S0": S0}
    meta = {"runs": []}
    for mode, pA in (("nmf", ("plus",)), ("mixmf", ("unity_plus", 1))):
        pS = ("plus",)
        prox = [spec_to_callable(pA), spec_to_callable(pS)]
        grad = partial(rnmf.grad_likelihood, Y=Y)
        f = partial(rnmf.log_likelihood, Y=Y)
        runs = [("pgm_bt", None)] + [("%s_%g" % (s, a), (s, a)) for a in (0.01, 0.1) for s in ("adam", "padam", "amsgrad")]
        for tag, cfg in runs:
            A, S = A0.copy(), S0.copy()
            tb = rutils.Traceback()
            if cfg is None:
                ralg.pgm([A, S], grad, rnmf.step_pgm, prox=prox, e_rel=1e-4, max_iter=1000, backtracking=True, f=f, callback=tb)
            else:
                sch, a = cfg
                ralg.adaprox([A, S], grad, lambda *X, it=None, a=a: (a, a), prox=prox, e_rel=1e-4, max_iter=1000,
                             scheme=sch, callback=tb)
            key = "%s/%s" % (mode, tag)
            blob[key + "/A"], blob[key + "/S"] = A, S
            meta["runs"].append({"key": key, "mode": mode, "prox_A": pA, "prox_S": pS, "cfg": cfg,
                                 "loss": float(f(A, S)), "iters": len(tb.trace)})
            print("  unmixing %-22s loss=%.10f its=%d" % (key, f(A, S), len(tb.trace)))
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "unmixing.npz"), **blob)
    print("wrote unmixing.npz")


def write_weighted_fixture():
    """Weighted likelihood W (M x N array; nmf.py:13-41) through nmf(): adaprox (its step rule ignores W) and pgm with a
    user step.  pgm's default step rule tests `W == 1` on the array (nmf.py:63) and raises: the exception type is recorded."""
    blob, meta = {}, {"cases": {}, "numpy": np.__version__, "reference": "proxmin 0.6.12"}
    for tag, (M, N, K, dtype) in {"f64": (40, 56, 4, np.float64), "f32": (64, 96, 8, np.float32)}.items():
        rng = np.random.default_rng(4321)
        Y, A0, S0 = problem(M, N, K, dtype, False, 4321)
        W = (0.2 + 1.8 * rng.random((M, N))).astype(dtype)
        W[rng.random((M, N)) < 0.1] = 0                      # masked entries
        blob[tag + "/Y"], blob[tag + "/A0"], blob[tag + "/S0"], blob[tag + "/W"] = Y, A0, S0, W
        blob[tag + "/loss0"] = rnmf.log_likelihood(A0, S0, Y=Y, W=W)
        gA, gS = rnmf.grad_likelihood(A0, S0, Y=Y, W=W)
        blob[tag + "/gA0"], blob[tag + "/gS0"] = gA, gS
        s_const = float(0.5 / max(np.linalg.eigvalsh(S0 @ S0.T)[-1], np.linalg.eigvalsh(A0.T @ A0)[-1]) / max(W.max(), 1.0))
        runs = {
            "amsgrad": dict(algorithm=ralg.adaprox, scheme="amsgrad", check_convergence=False),
            "adam_unityS": dict(algorithm=ralg.adaprox, scheme="adam", check_convergence=False,
                                prox_S=partial(rops.prox_unity_plus, axis=0)),
            "pgm_const_step": dict(algorithm=ralg.pgm, step=lambda *X, it=None: (s_const, s_const)),
        }
        for name, kw in runs.items():
            A, S = A0.copy(), S0.copy()
            tb = rutils.Traceback()
            rnmf.nmf(Y, A, S, W=W, max_iter=10, e_rel=1e-6, callback=tb, **kw)
            key = "%s/%s" % (tag, name)
            blob[key + "/A"], blob[key + "/S"] = A, S
            blob[key + "/loss"] = rnmf.log_likelihood(A, S, Y=Y, W=W)
            blob[key + "/n_callbacks"] = len(tb.trace)
            print("  weighted %-4s %-16s loss=%.9g its=%d" % (tag, name, blob[key + "/loss"], len(tb.trace)))
        meta["cases"][tag] = {"M": M, "N": N, "K": K, "s_const": s_const}
        try:
            rnmf.nmf(Y, A0.copy(), S0.copy(), W=W, max_iter=2)
            meta["cases"][tag]["default_step_error"] = None
        except Exception as e:   # noqa: BLE001
            meta["cases"][tag]["default_step_error"] = type(e).__name__
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "weighted.npz"), **blob)
    print("wrote weighted.npz", meta["cases"])


def write_array_step_fixture():
    """algorithms.pgm called directly with a user `step` that returns ARRAYS (they broadcast against the blocks:
    algorithms.py:106-108 multiplies S[j] into G[j] and hands S[j] to prox[j]): per-component vectors, full-shape and row
    arrays with an operator whose threshold scales with the step (prox_soft), plain and accelerated."""
    blob, meta = {}, {"cases": {}, "numpy": np.__version__, "reference": "proxmin 0.6.12"}
    for tag, (M, N, K, dtype) in {"f64": (33, 47, 3, np.float64), "f32": (64, 96, 8, np.float32)}.items():
        Y, A0, S0 = problem(M, N, K, dtype, False, 2468)
        rng = np.random.default_rng(1357)
        sA, sS = rnmf.step_pgm(A0, S0)
        vecA = (0.5 * sA * (1.0 + 0.1 * np.arange(K))).astype(dtype)                    # (K,)   against A (M, K)
        vecS = (0.5 * sS * (1.0 + 0.1 * np.arange(K)))[:, None].astype(dtype)           # (K, 1) against S (K, N)
        fullA = (0.5 * sA * (0.5 + rng.random((M, K)))).astype(dtype)                   # (M, K)
        rowS = (0.5 * sS * (0.5 + rng.random((1, N)))).astype(dtype)                    # (1, N)
        blob[tag + "/Y"], blob[tag + "/A0"], blob[tag + "/S0"] = Y, A0, S0
        blob[tag + "/vecA"], blob[tag + "/vecS"], blob[tag + "/fullA"], blob[tag + "/rowS"] = vecA, vecS, fullA, rowS
        soft = partial(rops.prox_soft_plus, thresh=0.05, type="relative")
        runs = {
            "vectors_plus": dict(step=(vecA, vecS), prox=[rops.prox_plus, rops.prox_plus], accelerated=False),
            "vectors_plus_fista": dict(step=(vecA, vecS), prox=[rops.prox_plus, rops.prox_plus], accelerated=True),
            "full_row_soft": dict(step=(fullA, rowS), prox=[soft, soft], accelerated=False),
            "scalar_and_vector": dict(step=(float(0.5 * sA), vecS), prox=[rops.prox_plus, soft], accelerated=True),
        }
        grad = partial(rnmf.grad_likelihood, Y=Y)
        for name, kw in runs.items():
            A, S = A0.copy(), S0.copy()
            tb = rutils.Traceback()
            st = kw["step"]
            conv, G, steps = ralg.pgm([A, S], grad, lambda *X, it=None, st=st: st, prox=kw["prox"], accelerated=kw["accelerated"],
                                      e_rel=1e-6, max_iter=10, callback=tb)
            key = "%s/%s" % (tag, name)
            blob[key + "/A"], blob[key + "/S"] = A, S
            blob[key + "/gA"], blob[key + "/gS"] = G
            blob[key + "/n_callbacks"] = len(tb.trace)
            print("  array steps %-4s %-20s its=%d  |A| %.6g |S| %.6g" % (tag, name, len(tb.trace), np.abs(A).sum(), np.abs(S).sum()))
        meta["cases"][tag] = {"M": M, "N": N, "K": K, "runs": sorted(runs)}
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "array_steps.npz"), **blob)
    print("wrote array_steps.npz")


def write_bt_user_step_fixture():
    """algorithms.pgm with backtracking=True AND a user `step` (algorithms.py:106 with :110-127): multiples of the Lipschitz
    steps, so that the line search has to halve; with and without `grads` in the callable's signature, plain and accelerated."""
    blob, meta = {}, {"cases": {}, "numpy": np.__version__, "reference": "proxmin 0.6.12"}
    for tag, (M, N, K, dtype) in {"f64": (33, 47, 3, np.float64), "f32": (64, 96, 8, np.float32)}.items():
        Y, A0, S0 = problem(M, N, K, dtype, False, 9753)
        blob[tag + "/Y"], blob[tag + "/A0"], blob[tag + "/S0"] = Y, A0, S0
        grad = partial(rnmf.grad_likelihood, Y=Y)
        f = partial(rnmf.log_likelihood, Y=Y)
        seen = {"grads": 0}

        def scaled(fac):
            def st(*X, it=None):
                return tuple(fac * s for s in rnmf.step_pgm(*X))
            return st

        def step2_grads(*X, it=None, grads=None):
            seen["grads"] += int(grads is not None)
            return tuple(2.0 * s for s in rnmf.step_pgm(*X))
        # 1.5 x and 2 x the Lipschitz steps: the search halves T and the runs stay monotone; larger factors (and any factor > 1
        # with acceleration) send the reference's own iteration off to 1e5 .. 1e8 -- nothing to compare trajectories on
        runs = {"x1.5": (scaled(1.5), False), "x2_with_grads": (step2_grads, False), "x1_fista": (scaled(1.0), True)}
        for name, (st, accel) in runs.items():
            A, S = A0.copy(), S0.copy()
            tb = rutils.Traceback()
            conv, G, steps = ralg.pgm([A, S], grad, st, prox=[rops.prox_plus, rops.prox_plus], accelerated=accel, backtracking=True, f=f,
                                      e_rel=1e-6, max_iter=15, callback=tb)
            key = "%s/%s" % (tag, name)
            blob[key + "/A"], blob[key + "/S"] = A, S
            blob[key + "/steps"] = np.array([float(steps[0]), float(steps[1])])
            blob[key + "/loss"] = rnmf.log_likelihood(A, S, Y=Y)
            blob[key + "/n_callbacks"] = len(tb.trace)
            print("  bt + user step %-4s %-10s its=%d loss=%.9g" % (tag, name, len(tb.trace), blob[key + "/loss"]))
        meta["cases"][tag] = {"M": M, "N": N, "K": K, "runs": sorted(runs)}
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "bt_user_step.npz"), **blob)
    print("wrote bt_user_step.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bt_user_step":
        write_bt_user_step_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "array_steps":      # one fixture only (the others are unchanged)
        write_array_step_fixture()
        sys.exit(0)
    all_names = list(CASES)
    print("nmf 200x1000 K=5 fp64 (SURVEY section 4 table; inputs regenerated from seed by the tests)")
    write_nmf_fixture("nmf_200x1000_k5_f64.npz", 200, 1000, 5, np.float64, 25, 1e-6, False,
                      ["pgm", "fista_half", "adam", "amsgrad", "amsgrad_unityS", "bsdmm_none", "bsdmm_plus_soft"], n_trace=0)
    print("nmf 33x47 K=3 fp64")
    write_nmf_fixture("nmf_33x47_k3_f64.npz", 33, 47, 3, np.float64, 12, 1e-6, True, all_names)
    print("nmf 64x96 K=8 fp32")
    # radam diverges to NaN in fp32 on this problem in the reference itself -> fp64 fixture only
    write_nmf_fixture("nmf_64x96_k8_f32.npz", 64, 96, 8, np.float32, 12, 1e-6, True, [n for n in all_names if n != "radam"])
    write_operator_fixture()
    write_helper_fixture()
    write_unmixing_fixture()
    write_weighted_fixture()
    write_array_step_fixture()
    write_bt_user_step_fixture()
