"""Solver back-ends of the nmf() path with the reference's signatures (proxmin/algorithms.py:
pgm :12-23, adaprox :248-265, bsdmm :653-666).  They are the objects `nmf()` compares `algorithm`
against by identity (nmf.py:141) and the host-side drivers of the device kernel chains.

Scope: the gradient must be the NMF likelihood gradient built by `proxmin_amd.nmf` (a
`functools.partial(nmf.grad_likelihood, Y=..., W=1)`) -- that is what carries Y to the device.
Constraints and step rules that are objects of this library (`proxmin_amd.operators.*`, bare,
`functools.partial` or `AlternatingProjections`; `nmf.step_pgm`, `nmf.step_adaprox`,
`nmf.scaled_step_pgm`, `nmf.constant_step`, `utils.BarzilaiBorweinStepper`) run fused inside the kernel
chains.  ANY OTHER Python callable keeps the reference's contract -- `prox(X, step) -> X'`
(algorithms.py:37-39), `step(*X, it=None[, grads=None])` (:73-77, :370) -- through a host round trip:
one iteration per call, the callable's arguments copied to the host ((M + N) K floats), its result
copied back, everything else (gradient, update, norms, built-in operators) still on the device.  A
one-time warning says so.  A user `grad(*X) -> (gA, gS)` callable (any differentiable function of the two factors,
algorithms.py:12,248) takes the same road: the point is copied to the host, the callable's result into the device's gradient
buffers, and the update runs on the device (pgm and adaprox; no Y is needed then).  bsdmm: user-defined members of `proxs_g`
and user-defined `prox_A` / `prox_S` are applied between the pieces of a block update (pmx_bsdmm_split); generic, untagged
`proxs_f` / `steps_f_cb` closures are the one thing left out (they would make the whole solver a host loop).
"""
from __future__ import annotations

import logging
from functools import partial

import numpy as np

from . import _lib, operators, utils
from .engine import DeviceNMF, open_weighted, as_device_array, DeviceArrayRef

logger = logging.getLogger("proxmin")


# ---------------------------------------------------------------------------------------------
def _is_nmf_grad(grad):
    from . import nmf as _nmf
    return isinstance(grad, partial) and grad.func is _nmf.grad_likelihood and not grad.args


def _problem_from_grad(X, grad):
    """-> (Y, A, S, W) for the NMF likelihood gradient built by proxmin_amd.nmf, (None, A, S, None) for any other callable
    (its result is copied into the device's gradient buffers once per iteration)."""
    from . import nmf as _nmf

    X = utils._as_tuple(X)
    assert len(X) == 2, "X must be [A, S]"
    A, S = X
    assert A.ndim == 2 and S.ndim == 2 and A.shape[1] == S.shape[0]
    if not _is_nmf_grad(grad):
        if not callable(grad):
            raise TypeError("grad must be callable")
        _warn_host_path("grad (%r)" % (grad,))
        return None, A, S, None
    kw = grad.keywords
    Y = as_device_array(kw["Y"])          # [r6] a Y that already lives in HBM (torch / CuPy) is adopted in place, never copied to the host
    if Y is None:
        Y = np.asarray(kw["Y"])
    assert tuple(Y.shape) == (A.shape[0], S.shape[1]), "Y must be M x N"
    return Y, A, S, _nmf._weights(kw.get("W", 1), Y.shape)


def _host_gradient(dev, grad, dt, accelerated_eval=True):
    """Evaluate a user `grad` at the device's current evaluation point and hand the result to the device."""
    Xe = (dev.get(_lib.BUF_EVAL_A, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_EVAL_A, 1)).astype(dt))
    G = utils._as_tuple(grad(*Xe))
    assert len(G) == 2 and np.shape(G[0]) == Xe[0].shape and np.shape(G[1]) == Xe[1].shape, "grad must return one array per block"
    dev.put(_lib.BUF_GA, 0, np.asarray(G[0]))
    dev.put(_lib.BUF_GA, 1, np.asarray(G[1]))
    return Xe


def _open_device(Y, A, S, W, f64=False, f64_mfma=False):
    """Context for one solver call; weights (nmf.py:13-41): engine.open_weighted picks the kernel.  f64: the caller's
    arrays are fp64 and the call is one the fp64 kernels cover (pgm / FISTA, small problem): compute in fp64 like the
    reference does for fp64 inputs (nmf.py:39-41)."""
    if f64:
        try:                            # (a weighted likelihood: the matrix-core kernels whatever the shape -- the small-problem ones take no weights)
            dev = DeviceNMF(A.shape[0], S.shape[1], A.shape[1], mode="f64" if (W is None and not f64_mfma) else "f64mfma", device=getattr(Y, "device", 0) if isinstance(Y, DeviceArrayRef) else 0)
        except NotImplementedError:     # the library's own test (pmx_ctx_create) is the authority: compute in fp32 and cast back, as for any other fp64 call
            dev = None
        if dev is not None:
            dev.set_Y(Y)
            if W is not None:
                dev.set_W(W)
            dev.set_factors(A, S)
            return dev
    if isinstance(Y, DeviceArrayRef) and Y.dtype == np.float64:
        raise NotImplementedError("a float64 device-resident Y is taken by the fp64 kernels only (library operators and step rules, no "
                                  "Barzilai-Borwein / user callables, K <= 128); hand this call a float32 array")
    _warn_f64_in_f32(Y, A, S)
    if Y is None:                       # a user `grad`: nothing M x N on the device (engine.DeviceNMF.set_host_grad)
        dev = DeviceNMF(A.shape[0], S.shape[1], A.shape[1], mode="f32")
        dev.set_host_grad(True)
        dev.set_factors(A, S)
        return dev
    dev = open_weighted(A.shape[0], S.shape[1], A.shape[1], W, device=getattr(Y, "device", 0) if isinstance(Y, DeviceArrayRef) else 0)
    dev.set_Y(Y)
    dev.set_factors(A, S)
    return dev


_f64_warned = []


def _warn_f64_in_f32(Y, A, S):
    """[r6] The reference computes in the dtype of the caller's arrays (nmf.py:39-41: NumPy keeps fp64).  fp64 arrays that the fp64
    kernels do not take (PMX_MODE_F64: all of the iteration on the device with library operators and step rules; PMX_F64_BIG=0) are computed in fp32 on the
    device and cast back: said ONCE, on logger "proxmin", so that nobody mistakes the result for an fp64 one."""
    if _f64_warned:
        return
    if any(getattr(x, "dtype", None) == np.float64 for x in (Y, A, S) if x is not None):
        _f64_warned.append(True)
        logger.warning("proxmin_amd: float64 arrays at %d x %d x %d are computed in float32 on the GPU and cast back (the fp64 kernels take "
                       "library operators and step rules only: no Barzilai-Borwein or user callables); the reference would keep float64 here. "
                       "Pass float32 arrays to silence this." % (A.shape[0], S.shape[1], A.shape[1]))


def _prox_pair(prox, n=2):
    prox = utils._as_tuple(prox)
    if len(prox) == 1:
        prox = prox * n
    assert len(prox) == n
    return prox


def _e_rel_pair(e_rel):
    if np.isscalar(e_rel):
        e_rel = (e_rel,) * 2
    assert len(e_rel) == 2
    return tuple(float(e) for e in e_rel)


def _write_back(dev, A, S):
    dA, dS = dev.get_factors()
    A[...] = dA
    S[...] = dS


def _wants_iterates(callback):
    return callback is not None and not isinstance(callback, utils.NullCallback)


_warned = set()


def _warn_host_path(what):
    if what not in _warned:
        _warned.add(what)
        logger.warning("proxmin_amd: %s is a user-defined Python callable: it is applied on the host once per iteration "
                       "(device -> host -> device copies of the factors); use the operators / step rules of this package to keep "
                       "the whole iteration on the GPU" % what)


def _split_prox(prox, none_is_id):
    """-> (device operator sequences, [callable or None per block]): operators of this library become device sequences,
    anything else is kept for the host round trip (its device slot is prox_id / no operator)."""
    seqs, host = [], []
    for j, p in enumerate(prox):
        q = operators.prox_id if (p is None and none_is_id) else p
        try:
            seqs.append(operators.device_proxseq(q, j, for_solver=True))
            host.append(None)
        except NotImplementedError as exc:
            if not callable(q):
                raise
            if isinstance(exc, operators.NotFusable):
                if "unity-long-%d" % j not in _warned:
                    _warned.add("unity-long-%d" % j)
                    logger.warning("proxmin_amd: %s: one iteration per call, its argument goes through the host" % exc)
            else:
                _warn_host_path("prox of block %d (%r)" % (j, q))
            seqs.append(operators.device_proxseq(operators.prox_id if none_is_id else None, j))
            host.append(q)
    return seqs, host


def _user_steps(s, what, shapes):
    """What a user `step` returned for pgm -> (one scalar per block for the device -- 1.0 stands in where the block's step is
    an array --, [None or the array per block], the values as the caller gets them back / as a host prox is handed them).
    Arrays must broadcast against their block (algorithms.py:106-108 multiplies them into the gradient)."""
    s = utils._as_tuple(s)
    if len(s) == 1:
        s = s * 2
    assert len(s) == 2, "%s must return one step per block" % what
    scal, arrs, orig = [], [], []
    for j, v in enumerate(s):
        a = np.asarray(v)
        if a.size == 1:
            scal.append(float(a.reshape(())))
            arrs.append(None)
            orig.append(float(a.reshape(())))
        else:
            np.broadcast_shapes(a.shape, shapes[j])      # raises ValueError like NumPy would inside the reference
            if np.broadcast_shapes(a.shape, shapes[j]) != tuple(shapes[j]):
                raise ValueError("step of block %d has shape %r: it does not broadcast against %r" % (j, a.shape, tuple(shapes[j])))
            scal.append(1.0)
            arrs.append(a)
            orig.append(v)
    return tuple(scal), arrs, orig


def _component_steps(alpha, K, j):
    """adaprox: a user step's Alpha[j] (scalar, or anything that broadcasts over the K components of block j like
    nmf.step_adaprox's (K,) for A and (K, 1) for S) -> K per-component values."""
    a = np.asarray(alpha, dtype=np.float64)
    if a.size == 1:
        return np.full(K, float(a.reshape(())), np.float32)
    want = (K,) if j == 0 else (K, 1)
    if a.shape in (want, (K,), (1, K) if j == 0 else (K, 1)) and a.size == K:
        return a.reshape(K).astype(np.float32)
    raise NotImplementedError("user `step` for adaprox must return scalars or per-component arrays (shape (K,) for A, (K, 1) for S); got %r" % (a.shape,))


# ---------------------------------------------------------------------------------------------
def pgm(X, grad, step, prox=None, accelerated=False, backtracking=False, f=None, e_rel=1e-6, max_iter=1000, callback=None):
    """Proximal Gradient Method / FISTA for the two NMF blocks (algorithms.py:12-144).

    Returns (converged, gradient, step) like the reference.  `step`: the default rule
    (`partial(nmf.step_pgm, W=1)` / `nmf.step_pgm`), `nmf.scaled_step_pgm(c)` or
    `nmf.constant_step(a, b)`.
    """
    from . import nmf as _nmf

    Y, A, S, W = _problem_from_grad(X, grad)
    prox = _prox_pair(prox)
    # prox=None means prox_id in pgm (algorithms.py:63-64)
    seqs, host_prox = _split_prox(prox, none_is_id=True)
    e_rel = _e_rel_pair(e_rel)
    user_grad = Y is None
    assert backtracking is False or f is not None
    if backtracking and user_grad:
        raise NotImplementedError("backtracking with a user-defined `grad` (its `f` would be evaluated on the host as well) is not implemented")
    if backtracking:
        # the smooth function must be the NMF likelihood of the same Y (it is evaluated by the fused residual kernel)
        ok = isinstance(f, partial) and f.func is _nmf.log_likelihood and not f.args and f.keywords.get("Y") is grad.keywords["Y"]
        if not ok:
            raise NotImplementedError("backtracking on the device needs f=functools.partial(proxmin_amd.nmf.log_likelihood, Y=Y) "
                                      "with the same Y as the gradient")
        fW = f.keywords.get("W", 1)
        if not (fW is grad.keywords.get("W", 1) or (np.isscalar(fW) and np.isscalar(grad.keywords.get("W", 1)) and fW == grad.keywords.get("W", 1))):
            raise NotImplementedError("backtracking: f and grad must carry the same weights W")
    scale, fixed, bb, user_step = 1.0, None, None, None
    bb_owner = getattr(step, "__self__", step)
    if isinstance(bb_owner, utils.BarzilaiBorweinStepper):
        bb = bb_owner
    elif isinstance(step, _nmf.scaled_step_pgm):
        scale = step.scale
    elif isinstance(step, _nmf.constant_step):
        fixed = step.steps
    elif step is _nmf.step_pgm or (isinstance(step, partial) and step.func is _nmf.step_pgm):
        # nmf.step_pgm tests `W == 1` on ITS OWN argument (nmf.py:63): an array there raises -- which is what nmf() hands it for a
        # weighted problem (nmf.py:152) --, the bare function (W = 1) gives the unweighted rule whatever weights the gradient carries
        sW = step.keywords.get("W", 1) if isinstance(step, partial) else 1
        if not np.isscalar(sW):
            raise ValueError(_nmf._AMBIGUOUS)
        if sW != 1:
            raise NotImplementedError("the weighted step rule of nmf.step_pgm (nmf.py:64-88) is not implemented; pass `step`")
    elif callable(step):
        user_step = step
        _warn_host_path("step (%r)" % (step,))
    else:
        raise TypeError("step must be callable")
    slow = user_step is not None or any(h is not None for h in host_prox) or user_grad
    # a user `step` next to the line search: the callable on the host once per iteration, the Beck-Teboulle loop on the device
    bt_user_step = backtracking and user_step is not None and not any(h is not None for h in host_prox) and not user_grad
    # [r4] a user `prox` inside the line search: every trial of THAT block takes a host round trip (pmx_pgm_bt_split), the
    # test itself, the other block and the likelihood evaluations stay on the device; a user `step` may come with it
    bt_user_prox = backtracking and any(h is not None for h in host_prox) and not user_grad and bb is None
    # Barzilai-Borwein steps next to a user-defined prox / grad: the rule stays on the device (it is the step), the callable
    # takes its round trip as with any other rule
    if slow and backtracking and not (bt_user_step or bt_user_prox):
        raise NotImplementedError("a user-defined grad (or Barzilai-Borwein steps with a user prox) together with backtracking is not implemented")

    # (scaled_step_pgm stands for `lambda *X, it=None: tuple(c * s for s in step_pgm(*X))`: step_pgm with ITS default W = 1 -- the
    # unweighted Lipschitz rule -- also when the likelihood carries weights)
    # [r4] fp64 inputs of a small problem, everything of the iteration on the device: fp64 arithmetic (PMX_MODE_F64)
    from .engine import f64_applies
    # [r6] ... at any size up to K = 128, weighted or not, with the Beck-Teboulle line search (the matrix-core kernels: k_big_f64.hip)
    f64 = (not slow and bb is None and Y is not None and not isinstance(W, DeviceArrayRef)
           and all(x.dtype == np.float64 for x in (Y, A, S)) and f64_applies(A.shape[0], S.shape[1], A.shape[1], weighted=W is not None or backtracking))
    with _open_device(Y, A, S, W, f64=f64, f64_mfma=backtracking) as dev:
        dev.pgm_begin(seqs, accelerated=accelerated, step_scale=scale, fixed_steps=(1.0, 1.0) if user_step is not None else fixed,
                      e_rel=e_rel, bb=(bb.type, bb.r) if bb is not None else None, backtracking=backtracking,
                      host_prox=[h is not None for h in host_prox], unweighted_rule=W is not None and user_step is None and fixed is None and bb is None)
        res = None
        it_done = 0
        dt = A.dtype
        steps_user, step_arrays = None, [None, None]
        if bt_user_prox:
            takes_grads = False
            if user_step is not None:                            # the reference's signature probe (algorithms.py:73-77)
                try:
                    user_step(A, S, it=0, grads=(A, S))
                    takes_grads = True
                except TypeError:
                    takes_grads = False
            for it in range(max_iter):
                if _wants_iterates(callback):
                    try:
                        callback(A, S, it=it)
                    except StopIteration:
                        break
                if user_step is not None:
                    if takes_grads:
                        dev.pgm_split(0)                         # the gradient at the evaluation point, for the callable only
                    Xe = (dev.get(_lib.BUF_EVAL_A, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_EVAL_A, 1)).astype(dt))
                    if takes_grads:
                        Gh = (dev.get(_lib.BUF_GA, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_GA, 1)).astype(dt))
                        ret = user_step(*Xe, it=it, grads=Gh)
                    else:
                        ret = user_step(*Xe, it=it)
                    steps, step_arrays, _ = _user_steps(ret, "step", (A.shape, S.shape))
                    if any(a is not None for a in step_arrays):
                        raise NotImplementedError("array-valued steps from a user `step` together with backtracking are not implemented")
                    dev.pgm_set_fixed_steps(steps)
                need, eff, res = dev.pgm_bt_split(0)
                while need:                                      # prox[j](_X[j] - T[j] S[j] G[j], T[j] S[j]) (algorithms.py:108, :125)
                    for j in range(2):
                        if (need >> j) & 1:
                            T = np.ascontiguousarray(dev.get(_lib.BUF_BT_A, j)).astype(dt)
                            dev.put(_lib.BUF_BT_A, j, np.asarray(host_prox[j](T, dt.type(eff[j]))))
                    need, eff, res = dev.pgm_bt_split(1)
                _write_back(dev, A, S)
                it_done = res.total_iterations
                if res.stopped:
                    break
        elif bt_user_step:
            # algorithms.py:105-127 with a Python `step`: its scalars become the constants of ONE device iteration with the
            # line search (T[j] S[j] in the reference: T lives on the device, S comes from here)
            takes_grads = False
            try:                                                # the reference's signature probe (algorithms.py:73-77)
                user_step(A, S, it=0, grads=(A, S))
                takes_grads = True
            except TypeError:
                takes_grads = False
            for it in range(max_iter):
                if _wants_iterates(callback):
                    try:
                        callback(A, S, it=it)
                    except StopIteration:
                        break
                if takes_grads:
                    dev.pgm_split(0)                            # the gradient at the evaluation point, for the callable only
                Xe = (dev.get(_lib.BUF_EVAL_A, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_EVAL_A, 1)).astype(dt))
                if takes_grads:
                    Gh = (dev.get(_lib.BUF_GA, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_GA, 1)).astype(dt))
                    ret = user_step(*Xe, it=it, grads=Gh)
                else:
                    ret = user_step(*Xe, it=it)
                steps, step_arrays, _ = _user_steps(ret, "step", (A.shape, S.shape))
                if any(a is not None for a in step_arrays):
                    raise NotImplementedError("array-valued steps from a user `step` together with backtracking are not implemented")
                dev.pgm_set_fixed_steps(steps)
                res = dev.pgm_run(1)
                _write_back(dev, A, S)
                it_done = res.total_iterations
                if res.stopped:
                    break
        elif slow:
            # One iteration per pass, in pieces (algorithms.py:87-135): the gradient at the (extrapolated) point on the
            # device, the user's step / prox on the host with exactly the arguments the reference passes, the update,
            # extrapolation and stopping test on the device again.
            takes_grads = False
            arrays_on, steps_user, step_arrays = False, None, [None, None]
            if user_step is not None:                           # the reference's signature probe (algorithms.py:73-77)
                try:
                    user_step(A, S, it=0, grads=(A, S))
                    takes_grads = True
                except TypeError:
                    takes_grads = False
            for it in range(max_iter):
                if _wants_iterates(callback):
                    try:
                        callback(A, S, it=it)
                    except StopIteration:
                        break
                if user_grad:                                   # grads = grad(*_X) (algorithms.py:105)
                    _host_gradient(dev, grad, dt)
                r0 = dev.pgm_split(0)
                steps = None
                if user_step is not None:
                    Xe = (dev.get(_lib.BUF_EVAL_A, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_EVAL_A, 1)).astype(dt))
                    if takes_grads:
                        Gh = (dev.get(_lib.BUF_GA, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_GA, 1)).astype(dt))
                        ret = user_step(*Xe, it=it, grads=Gh)
                    else:
                        ret = user_step(*Xe, it=it)
                    # scalars per block, or arrays that broadcast against the blocks (uploaded element by element)
                    steps, step_arrays, steps_user = _user_steps(ret, "step", (A.shape, S.shape))
                    if any(a is not None for a in step_arrays) or arrays_on:
                        dev.pgm_step_arrays(step_arrays)
                        arrays_on = any(a is not None for a in step_arrays)
                if any(h is not None for h in host_prox):
                    dev.pgm_split(1, steps)
                    for j, h in enumerate(host_prox):
                        if h is None:
                            continue
                        T = np.ascontiguousarray(dev.get(_lib.BUF_TMP_A, j)).astype(dt)
                        if steps is not None and step_arrays[j] is not None:
                            sj = np.asarray(steps_user[j]).astype(dt)        # the array, as the reference hands it on
                        else:
                            sj = dt.type(steps[j] if steps is not None else r0.steps[j])
                        out = h(T, sj)                           # prox(X, step) -> X' (algorithms.py:37-39, :108)
                        dev.put(_lib.BUF_TMP_A, j, np.asarray(out))
                res = dev.pgm_split(2, steps)
                _write_back(dev, A, S)
                it_done = res.total_iterations
                if res.stopped:
                    break
        elif _wants_iterates(callback):
            for it in range(max_iter):
                try:
                    callback(A, S, it=it)                       # algorithms.py:90 (pre-update iterate)
                except StopIteration:
                    break
                res = dev.pgm_run(1)
                _write_back(dev, A, S)
                it_done = res.total_iterations
                if res.stopped:
                    break
        else:
            res = dev.pgm_run(max_iter)
            it_done = res.total_iterations
            _write_back(dev, A, S)
        G = (dev.get(_lib.BUF_GA, 0).astype(dt), np.ascontiguousarray(dev.get(_lib.BUF_GA, 1)).astype(dt))
    converged = tuple(bool(c) for c in res.converged) if res is not None else (False, False)
    steps = (dt.type(res.steps[0]), dt.type(res.steps[1])) if res is not None else (None, None)
    if res is not None and slow and user_step is not None and steps_user is not None:
        steps = tuple(steps_user[j] if step_arrays[j] is not None else steps[j] for j in range(2))   # arrays go back as returned
    logger.info("Completed {0} iterations".format(it_done))
    if not all(converged):
        logger.warning("Solution did not converge")
    return converged, G, steps


# ---------------------------------------------------------------------------------------------
def adaprox(X, grad, step, prox=None, scheme="adam", b1=0.9, b2=0.999, eps=1e-8, check_convergence=True, p=0.25,
            e_rel=1e-6, max_iter=1000, prox_max_iter=1000, M=None, V=None, Vhat=None, callback=None):
    """Adaptive proximal gradient method, Adam family (algorithms.py:248-423).

    Returns (converged, M, V, Vhat).  As in the reference, a cold start returns Vhat = [None, None]
    even for amsgrad/padam/adamx (the running maximum is only kept when Vhat arrays are passed in).
    """
    from . import nmf as _nmf

    Y, A, S, W = _problem_from_grad(X, grad)
    prox = _prox_pair(prox)
    seqs, host_prox = _split_prox(prox, none_is_id=False)       # prox=None: no proximal loop at all (algorithms.py:380)
    e_rel = _e_rel_pair(e_rel)
    if not hasattr(b1, "__iter__"):
        b1 = np.array((b1,) * max_iter)
    b1 = np.asarray(b1, dtype=np.float64)
    assert len(b1) == max_iter
    assert (b1 >= 0).all() and (b1 < 1).all()
    assert b2 >= 0 and b2 < 1
    assert eps >= 0
    assert p > 0 and p <= 0.5
    scheme = scheme.lower()
    assert scheme in ["adam", "nadam", "adamx", "amsgrad", "padam", "radam"]
    fixed, user_step = None, None
    if isinstance(step, _nmf.constant_step):
        fixed = step.steps
    elif step is not _nmf.step_adaprox:
        if not callable(step):
            raise TypeError("step must be callable")
        user_step = step
        _warn_host_path("step (%r)" % (step,))
    user_grad = Y is None
    slow = user_step is not None or any(h is not None for h in host_prox) or user_grad

    Xs = (A, S)
    warm = M is not None or V is not None
    if M is not None:
        assert len(M) == 2 and all(m.shape == x.shape for x, m in zip(Xs, M))
    if V is not None:
        assert len(V) == 2 and all(v.shape == x.shape for x, v in zip(Xs, V))
    if Vhat is not None:
        assert len(Vhat) == 2 and all(vh.shape == x.shape for x, vh in zip(Xs, Vhat))

    # [r4] fp64 inputs of a small problem: fp64 arithmetic on the device (k_small_f64.hip: k64_ada_iter), as the reference's own
    from .engine import f64_applies
    f64 = (not slow and Y is not None and not isinstance(W, DeviceArrayRef) and all(x.dtype == np.float64 for x in (Y, A, S))
           and f64_applies(A.shape[0], S.shape[1], A.shape[1], weighted=W is not None))
    with _open_device(Y, A, S, W, f64=f64) as dev:
        for j in range(2):
            if warm:
                dev.put(_lib.BUF_MA, j, M[j] if M is not None else np.zeros(Xs[j].shape, np.float32))
                dev.put(_lib.BUF_VA, j, V[j] if V is not None else np.zeros(Xs[j].shape, np.float32))
            if Vhat is not None:
                dev.put(_lib.BUF_VHA, j, Vhat[j])
        dev.adaprox_begin(seqs, scheme=scheme, b2=b2, eps=eps, p=p, check_convergence=check_convergence,
                          prox_max_iter=prox_max_iter, warm_moments=warm, warm_vhat=Vhat is not None,
                          fixed_alpha=fixed, e_rel=e_rel, host_step=user_step is not None,
                          host_prox=[h is not None for h in host_prox])
        res = None
        it_done = 0
        host_sub = [0, 0]
        if max_iter > 0 and slow:
            # One iteration per pass (algorithms.py:365-410): user step -> device moments and update -> the proximal loop
            # of a block with a user-defined prox around that callable on the host (:383-400, with the reference's own
            # expressions) -> the other block's loop, X <- z, stopping test and next step sizes on the device.
            dt = A.dtype
            K = A.shape[1]
            for it in range(max_iter):
                if _wants_iterates(callback):
                    try:
                        callback(A, S, it=it)
                    except StopIteration:
                        break
                alpha = None
                if user_grad:                                                      # G = grad(*X) (algorithms.py:369)
                    _host_gradient(dev, grad, dt)
                if user_step is not None:
                    alpha = utils._as_tuple(user_step(A, S, it=it))                  # algorithms.py:370
                    assert len(alpha) == 2, "step must return one Alpha per block"
                    dev.adaprox_set_alpha(_component_steps(alpha[0], K, 0), _component_steps(alpha[1], K, 1))
                elif fixed is not None:
                    # nmf.constant_step: the device keeps the two constants (adaprox_begin(fixed_alpha=...)); recomputing the
                    # default rule here would overwrite DevStatus::alpha with mean(X)/10 (and clear the halt flag mid-run)
                    alpha = (dt.type(fixed[0]), dt.type(fixed[1]))
                elif any(h is not None for h in host_prox):
                    aA, aS = dev.step_adaprox()                                    # the rule the device applies (nmf.py:93)
                    alpha = (aA.astype(dt), aS.astype(dt)[:, None])
                if not any(h is not None for h in host_prox):                      # only the step rule is the user's
                    res = dev.adaprox_run(b1[it:it + 1], b1[it - 1])
                    _write_back(dev, A, S)
                    it_done = res.total_iterations
                    if res.stopped:
                        break
                    continue
                _, maxpsi = dev.adaprox_split(0, it, b1[it], b1[it - 1])
                taus = [0, 0]
                for j, h in enumerate(host_prox):
                    if h is None:
                        continue
                    Xj = np.ascontiguousarray(dev.get(_lib.BUF_A, j)).astype(dt)
                    Psi = np.ascontiguousarray(dev.get(_lib.BUF_PSI_A, j)).astype(dt)
                    Alpha = alpha[j] if not np.isscalar(alpha[j]) else dt.type(alpha[j])
                    z = Xj.copy()
                    gamma = Alpha / dt.type(maxpsi[j])                              # algorithms.py:384
                    tau = 0
                    for tau in range(1, prox_max_iter + 1):                        # :386-393
                        z_ = h(z - gamma / Alpha * Psi * (z - Xj), gamma)
                        converged_ = utils.l2sq(z_ - z) <= e_rel[j] ** 2 * utils.l2sq(z)
                        z = z_
                        if converged_:
                            break
                    dev.put(_lib.BUF_A, j, z)                                      # X[j][:] = z (:400)
                    taus[j] = tau
                    host_sub[j] += tau
                res, _ = dev.adaprox_split(1, it, b1[it], b1[it - 1], taus)
                _write_back(dev, A, S)
                it_done = res.total_iterations
                if res.stopped:
                    break
        elif max_iter > 0:
            if _wants_iterates(callback):
                for it in range(max_iter):
                    try:
                        callback(A, S, it=it)                   # algorithms.py:368
                    except StopIteration:
                        break
                    res = dev.adaprox_run(b1[it:it + 1], b1[it - 1])   # b1[-1] at it = 0, like algorithms.py:213
                    _write_back(dev, A, S)
                    it_done = res.total_iterations
                    if res.stopped:
                        break
            else:
                res = dev.adaprox_run(b1, b1[-1])
                it_done = res.total_iterations
                _write_back(dev, A, S)
        dt = A.dtype
        outM = tuple(np.ascontiguousarray(dev.get(_lib.BUF_MA, j)).astype(dt) for j in range(2))
        outV = tuple(np.ascontiguousarray(dev.get(_lib.BUF_VA, j)).astype(dt) for j in range(2))
        if M is not None:
            for j in range(2):
                M[j][...] = outM[j]
            outM = M
        if V is not None:
            for j in range(2):
                V[j][...] = outV[j]
            outV = V
        if Vhat is not None:
            for j in range(2):
                Vhat[j][...] = dev.get(_lib.BUF_VHA, j)
            outVhat = Vhat
        else:
            outVhat = [None] * 2
    sub = [int(res.sub_iterations[0]), int(res.sub_iterations[1])] if res is not None else [0, 0]
    logger.info("Completed {0} iterations and {1} sub-iterations".format(it_done, sub))
    if check_convergence:
        converged = tuple(bool(c) for c in res.converged) if res is not None else (False, False)
        if not all(converged):
            logger.warning("Solution did not converge")
    else:
        converged = (None,) * 2
    return converged, outM, outV, outVhat


# ---------------------------------------------------------------------------------------------
def bsdmm(X, proxs_f, steps_f_cb, proxs_g=None, steps_g=None, Ls=None, update_order=None, steps_g_update="steps_f",
          max_iter=1000, e_rel=1e-6, e_abs=0, callback=None):
    """Block-Simultaneous Direction Method of Multipliers (algorithms.py:653-850).

    `proxs_f(X, step, j=, Xs=)` and `steps_f_cb(Xs, j=)` must be the closures `nmf.bsdmm_closures(Y, prox)` builds --
    the ones the reference's nmf() builds inline (nmf.py:181-193): prox_j(X - step grad_j(Xs), step) and step_pgm(Xs)[j] --
    because they are what carries Y and the operators to the device.  Generic closures cannot be mapped to kernels.
    """
    tag_f, tag_s = getattr(proxs_f, "_pmx_nmf", None), getattr(steps_f_cb, "_pmx_nmf", None)
    if tag_f is None or tag_s is None or tag_f is not tag_s:
        # Generic closures (algorithms.py:805-821): nothing about the smooth function is known to the device.  It keeps what
        # IS generic -- dX from the constraint variables, the Z / U updates with the library's proxs_g members, the norms and
        # Boyd's test -- and the two closures are called on the host once per block update (pmx_bsdmm_split with a zero
        # device gradient: phase 0 hands out X_j - dX, the closure's result becomes the new X_j).
        return _bsdmm_nmf(X, None, (None, None), proxs_g=proxs_g, steps_g=steps_g, Ls=Ls, update_order=update_order,
                          steps_g_update=steps_g_update, max_iter=max_iter, e_rel=e_rel, e_abs=e_abs, callback=callback,
                          closures=(proxs_f, steps_f_cb))
    grad, prox = tag_f
    return _bsdmm_nmf(X, grad, prox, proxs_g=proxs_g, steps_g=steps_g, Ls=Ls, update_order=update_order,
                      steps_g_update=steps_g_update, max_iter=max_iter, e_rel=e_rel, e_abs=e_abs, callback=callback)


def _bsdmm_nmf(X, grad, prox, proxs_g=None, steps_g=None, Ls=None, update_order=None, steps_g_update="steps_f",
               max_iter=1000, e_rel=1e-6, e_abs=0, callback=None, closures=None):
    """The bsdmm branch of nmf() (nmf.py:178-203 -> algorithms.py:653-850): step_f = step_pgm,
    identity linear operators, steps_g from steps_f (algorithms.py:815-819)."""
    if closures is not None:
        _warn_host_path("proxs_f / steps_f_cb (%r, %r)" % closures)
        Xt = utils._as_tuple(X)
        if (len(Xt) != 2 or np.ndim(Xt[0]) != 2 or np.ndim(Xt[1]) != 2 or np.shape(Xt[0])[1] != np.shape(Xt[1])[0]):
            raise NotImplementedError("bsdmm with generic proxs_f / steps_f_cb closures runs on the device only for a two-block problem "
                                      "X = [A (M x K), S (K x N)] (got blocks of shapes %s); other block structures are outside the "
                                      "NMF/CMF path this library implements" % ([np.shape(x) for x in Xt],))
        Y, A, S, W = None, Xt[0], Xt[1], None
    else:
        Y, A, S, W = _problem_from_grad(X, grad)
    if W is not None:                # bsdmm's steps come from nmf.step_pgm, which raises on an array W (nmf.py:63,187-193)
        from . import nmf as _nmf_w
        raise ValueError(_nmf_w._AMBIGUOUS)
    N = 2
    if proxs_g is None:
        proxs_g = [None] * N
    assert len(proxs_g) == N
    steps_g_update = steps_g_update.lower()
    assert steps_g_update in ["steps_f", "fixed", "relative"]
    if steps_g_update != "steps_f" and steps_g is not None:
        # not reachable in the reference either: its per-iteration container steps_g_ is only ever filled by the "steps_f"
        # strategy (algorithms.py:773-781, :815-819), "fixed" hands update_variables a list holding None (TypeError) and
        # "relative" divides by the initial steps_f = None (:808-810)
        raise NotImplementedError("steps_g_update='%s' with explicit steps_g raises TypeError in the reference (algorithms.py:773-819) "
                                  "and is not implemented" % steps_g_update)
    if Ls is not None and any(L is not None for L in np.ravel(np.array(Ls, dtype=object))):
        raise NotImplementedError("linear operators Ls are not implemented on the device (identity only)")
    order = None if update_order is None else [int(j) for j in update_order]
    if order is not None:
        assert all(j in (0, 1) for j in order), "update_order refers to a block that does not exist"
        if len(order) == 0:
            # the reference loops max_iter times over an empty block list (algorithms.py:800-846): nothing is updated, the
            # callback still sees every iteration, `converged` keeps its initial [None, None] and the warning is logged.
            # (n_order = 0 means "default order" to the C ABI, so this case never reaches the device.)
            if _wants_iterates(callback):
                for it in range(max_iter):
                    callback(A, S, it=it)
            logger.info("Completed {0} iterations".format(max_iter))
            logger.warning("Solution did not converge")
            return [None, None]
    er = [e_rel] * N if np.isscalar(e_rel) else list(e_rel)
    ea = [e_abs] * N if np.isscalar(e_abs) else list(e_abs)
    def _seq_or_host(q, j, what):
        """device sequence of an operator of this library; (prox_id, callable) for anything else (host round trip)"""
        q = q if q is not None else operators.prox_id
        try:
            return operators.device_proxseq(q, j, for_solver=True), None
        except NotImplementedError as exc:
            if not callable(q):
                raise
            if isinstance(exc, operators.NotFusable):
                # [r4] this library's prox_unity* along the block's LONG axis: its grid-wide sum exists as a stand-alone device
                # kernel only, so the operator takes the same between-launches path a user callable takes (pmx_bsdmm_split) --
                # calling it on the host copy runs that kernel; no host arithmetic
                if "unity-long-bsdmm-%d" % j not in _warned:
                    _warned.add("unity-long-bsdmm-%d" % j)
                    logger.warning("proxmin_amd: %s: one iteration per call, its argument goes through the host" % exc)
            else:
                _warn_host_path("%s of block %d (%r)" % (what, j, q))
            return operators.device_proxseq(operators.prox_id, j), q

    seq_f, host_f = [], []
    for j, q in enumerate(prox):
        sq, h = _seq_or_host(q, j, "prox")
        seq_f.append(sq)
        host_f.append(h)
    seq_g, host_g = [], []
    for j in range(N):
        g = proxs_g[j]
        if g is None:
            seq_g.append(None)
            host_g.append([])
            continue
        if not hasattr(g, "__iter__"):
            g = [g]
        pairs = [_seq_or_host(q, j, "proxs_g member") for q in g]
        seq_g.append([sq for sq, _ in pairs])
        host_g.append([h for _, h in pairs])
    slow = any(h is not None for h in host_f) or any(h is not None for hs in host_g for h in hs) or closures is not None

    # [r4] fp64 inputs of a small problem: fp64 arithmetic on the device (k_small_f64.hip: k64_bsdmm_block), as the reference's own
    from .engine import f64_applies
    f64 = (not slow and Y is not None and all(x.dtype == np.float64 for x in (Y, A, S))
           and f64_applies(A.shape[0], S.shape[1], A.shape[1]))
    with _open_device(Y, A, S, None, f64=f64) as dev:
        dev.bsdmm_begin(seq_f, seq_g, e_rel=er, e_abs=ea, update_order=order)
        res = None
        if closures is not None:        # the device's gradient stays zero: X_j - dX comes out of phase 0
            dev.put(_lib.BUF_GA, 0, np.zeros(A.shape, np.float32))
            dev.put(_lib.BUF_GA, 1, np.zeros(S.shape, np.float32))
        if slow:
            # One iteration per pass, every block update in pieces (utils.py:307-346): step_f and the gradient on the device,
            # the user's prox_f on the host, X / dual updates and norms on the device, the user-defined members of proxs_g on
            # the host with exactly the arguments the reference passes (X + U_i, step_g_i), Boyd's test on the device.
            dt = A.dtype
            blocks = order if order is not None else [0, 1]
            for it in range(max_iter):
                if _wants_iterates(callback):
                    callback(A, S, it=it)                       # no StopIteration handler (algorithms.py:802)
                for o, j in enumerate(blocks):
                    hf = host_f[j] is not None or closures is not None
                    mask = sum(1 << i for i, h in enumerate(host_g[j]) if h is not None)
                    user_sf = float(closures[1]((A, S), j=j)) if closures is not None else 0.0   # steps_f_cb(X, j=j) (algorithms.py:807)
                    if closures is not None and not user_sf > 0.0:
                        raise ValueError("steps_f_cb returned %r for block %d: a positive step is required" % (user_sf, j))
                    r0 = dev.bsdmm_split(j, 0, hf, mask, step_f=user_sf)
                    step_f = dt.type(r0.steps[j])
                    rows = A.shape[0] if j == 0 else S.shape[1]
                    if hf:
                        T = np.ascontiguousarray(dev.get(_lib.BUF_TMP_A, j)).astype(dt)
                        if closures is not None:                # proxs_f(X_j - dX, step_f, j=j, Xs=X) (algorithms.py:806, utils.py:338)
                            out = closures[0](T, step_f, j=j, Xs=(A, S))
                        else:                                   # prox_j(X - dX - step grad_j, step) (nmf.py:181-185)
                            out = host_f[j](T, step_f)
                        dev.put(_lib.BUF_TMP_A, j, np.asarray(out))
                    dev.bsdmm_split(j, 1, hf, mask)
                    if mask:
                        step_g = dt.type(float(r0.steps[j]) * 2.0 * len(host_g[j]))           # utils.get_step_g, identity L (utils.py:269-279)
                        for i, h in enumerate(host_g[j]):
                            if h is None:
                                continue
                            buf = _lib.BUF_TG0 + j * _lib.MAX_G + i
                            T = dev._download(buf, rows)
                            T = np.ascontiguousarray(T if j == 0 else T.T).astype(dt)
                            out = np.asarray(h(T, step_g))                                   # Z_i = prox_g_i(L X + U_i, step_g_i) (utils.py:295-304)
                            dev._upload(buf, out if j == 0 else out.T)
                    res = dev.bsdmm_split(j, 2, hf, mask, last_block=(o == len(blocks) - 1))
                    if closures is not None:                    # Gauss-Seidel: the next block's closures see this update
                        _write_back(dev, A, S)
                _write_back(dev, A, S)
                if res is not None and res.stopped:
                    break
        elif _wants_iterates(callback):
            for it in range(max_iter):
                callback(A, S, it=it)                           # no StopIteration handler (algorithms.py:802)
                res = dev.bsdmm_run(1)
                _write_back(dev, A, S)
                if res.stopped:
                    break
        else:
            res = dev.bsdmm_run(max_iter)
            _write_back(dev, A, S)
    converged = [bool(c) for c in res.converged] if res is not None else [None, None]
    if order is not None:                # a block that is never updated keeps the reference's initial None (algorithms.py:794)
        converged = [converged[j] if j in order else None for j in range(2)]
    logger.info("Completed {0} iterations".format(res.total_iterations if res is not None else 0))
    if not all(converged):
        logger.warning("Solution did not converge")
    return converged
