"""Thin object wrapper around one libpmx context: device residency of Y / A / S^T for the
duration of an `nmf()` call, uploads/downloads with the S <-> S^T transposition, and the solver
entry points.  Pure plumbing -- every number is produced by the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
import logging
import os

import numpy as np

from . import _lib

logger = logging.getLogger("proxmin")
_noticed = set()


def _notice(key, msg):
    """one line per distinct situation on logger "proxmin" (INFO): which K1 a context really got"""
    if key not in _noticed:
        _noticed.add(key)
        logger.info(msg)


# Arithmetic of the three contractions inside the fused residual-gradient kernel:
#   "f32"    exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
#   "bf16x3" split-bf16 MFMA with fp32-class accuracy (3-term split for A@S, 2-term for the gradients);
#            K <= 64, larger K runs the fp32 kernel
#   "f16x2"  as "bf16x3", but shapes with K = 64, M % 128 = 0, N % 256 = 0 run the two-term fp16 kernel (operands scaled
#            by powers of two from the factor maxima: 9 instead of 12 MFMA products per multiply-add), and so do shapes
#            with K = 128, M % 128 = 0, N % 128 = 0 (k_grad_f16_k128; no weights there)
#   "f16x2r" "f16x2" with the RESIDUAL in exact fp32's class (include/pmx.h: PMX_MODE_F16X2R).  K1's K = 64 / 128 without weights: the
#            residual from the high x high fp16 product alone, the rest restored exactly through K x K matrices (k_gfix.hip) -- 7 instead
#            of 9 MFMA products per multiply-add; K = 32: third fp16 terms of A and S in a second accumulator (11 products)
# [r6] The library's default IS the benchmarked arithmetic: f16x2r is in exact fp32's error class, falls back to the
# exact-fp32 kernel by itself where it has no kernel of its own (off-shape weighted contexts: open_weighted) or where one fp16 scale cannot
# carry the residual (the range guard, tests/test_gpu_range.py), and runs the tuned kernels on zero-padded frames for ragged shapes.
LIBRARY_DEFAULT_MODE = "f16x2r"
_DEFAULT_MODE = os.environ.get("PMX_MODE", LIBRARY_DEFAULT_MODE)
assert _DEFAULT_MODE in ("f32", "bf16x3", "f16x2", "f16x2r"), "PMX_MODE must be one of f32, bf16x3, f16x2, f16x2r"


def set_default_mode(mode=None):
    """Select the contraction arithmetic used by nmf() and friends ("f32", "bf16x3", "f16x2" or "f16x2r"); None: back to the
    library's default (LIBRARY_DEFAULT_MODE, or PMX_MODE from the environment)."""
    global _DEFAULT_MODE
    if mode is None:
        mode = os.environ.get("PMX_MODE", LIBRARY_DEFAULT_MODE)
    assert mode in ("f32", "bf16x3", "f16x2", "f16x2r")
    _DEFAULT_MODE = mode


def get_default_mode():
    return _DEFAULT_MODE


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64_applies(M, N, K, weighted=False):
    """Shapes the fp64 kernels take (include/pmx.h: PMX_MODE_F64 -- the fused small-problem kernels for the reference's own examples
    and BASELINE cfg1, the MFMA passes of k_big_f64.hip for the rest up to K = 128); PMX_F64=0 switches the mode off (fp64 inputs are
    then computed in fp32 and cast back, with a warning)."""
    def off(name):                      # the library reads these with atoi(): anything that is not a non-zero number switches off
        v = os.environ.get(name)
        if v is None:
            return False
        try:
            return int(v.strip() or 0) == 0
        except ValueError:
            return True
    if off("PMX_F64"):
        return False
    small = not weighted and not off("PMX_K1_SMALL") and K <= 16 and M <= 4096 and N <= 8192 and M * N <= (1 << 20)
    # [r6] everything else up to K = 128 -- and every weighted likelihood -- : one MFMA pass per gradient (k_big_f64.hip);
    # PMX_F64_BIG=0 keeps the small kernels only
    return small or (not off("PMX_F64_BIG") and K <= 128 and 8.0 * M * N * (2 if weighted else 1) <= 160e9)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceArrayRef:
    """[r6] Y (or W) that is ALREADY in HBM: anything that speaks `__cuda_array_interface__` (a torch tensor on the GPU, a
    CuPy array): float32 (or float64: [r6] copied into the fp64 context's own array), row-major, unit stride along the rows.  nmf() and the solvers adopt float32 zero-copy (pmx_set_Y_device)
    instead of uploading a host array -- the reference's Y lives in host RAM (nmf.py:96), ours may live where the kernels read it.
    The owner must keep its memory alive and unchanged for the duration of the call (the context holds a reference)."""

    def __init__(self, obj):
        cai = obj.__cuda_array_interface__
        ts = cai.get("typestr")
        if ts not in ("<f4", "=f4", "|f4", "<f8", "=f8", "|f8"):
            raise TypeError("a device-resident Y must be float32 or float64 (got %r)" % (ts,))
        es = 8 if ts.endswith("8") else 4
        shape = tuple(int(v) for v in cai["shape"])
        if len(shape) != 2:
            raise TypeError("a device-resident Y must be two-dimensional")
        strides = cai.get("strides")
        if strides is None:
            ld = shape[1]
        else:
            if int(strides[1]) != es or int(strides[0]) % es or int(strides[0]) < es * shape[1]:
                raise TypeError("a device-resident Y must be row-major with unit stride along its rows")
            ld = int(strides[0]) // es
        self.ptr, self.shape, self.ld, self.owner = int(cai["data"][0]), shape, ld, obj
        self.dtype = np.dtype(np.float64 if es == 8 else np.float32)
        self.ndim = 2
        dev = getattr(obj, "device", None)
        self.device = int(getattr(dev, "index", None) or 0) if dev is not None and not isinstance(dev, int) else int(dev or 0)

    def __array__(self, *a, **k):
        raise TypeError("this array lives on the GPU; proxmin_amd adopts it in place")


def as_device_array(obj):
    """DeviceArrayRef for an object that lives on the GPU, None for host data"""
    if isinstance(obj, DeviceArrayRef):
        return obj
    if isinstance(obj, np.ndarray) or not hasattr(obj, "__cuda_array_interface__"):
        return None
    if getattr(obj, "is_cuda", True) is False:       # a torch tensor on the host
        return None
    return DeviceArrayRef(obj)


class DeviceNMF:
    """Device state for one factorisation problem Y (M x N) ~ A (M x K) @ S (K x N)."""

    def __init__(self, M, N, K, device=0, mode=None, stream=None):
        self.lib = _lib.require_gpu()
        self.M, self.N, self.K = int(M), int(N), int(K)
        self.device = device
        mode = mode or _DEFAULT_MODE
        self.mode = mode
        self.f64 = mode in ("f64", "f64mfma")          # fp64 operands, products and sums (small problems, the fused loops of the three back-ends: k_small_f64.hip)
        mode_id = {"f32": _lib.MODE_F32, "bf16x3": _lib.MODE_BF16X3, "f16x2": _lib.MODE_F16X2, "f16x2r": _lib.MODE_F16X2R, "f64": _lib.MODE_F64, "f64mfma": _lib.MODE_F64_MFMA}[mode]
        h = C.c_void_p()
        _lib.check(self.lib.pmx_ctx_create(C.byref(h), device, self.M, self.N, self.K, mode_id,
                                           C.c_void_p(stream) if stream else None))
        self.h = h
        self._keep = []
        # a split-precision context that fell off its fast kernel says so once (pmx_k1_info knows): ragged shapes and
        # K outside {64, 128} run the generic split-bf16 kernels or the exact-fp32 one, 1.5-3 x slower per pass
        if mode not in ("f32", "f64", "f64mfma"):
            k = self.k1_info()["kernel"]
            fast = {"f16x2": ("k_grad_f16_v8", "k_grad_f16_k128", "k_grad_f16_k32", "k_grad_small"), "bf16x3": ("k_grad_bf16", "k_grad_small"),
                    "f16x2r": ("k_grad_f16_v8_r3", "k_grad_f16_k32_r3", "k_grad_f16_v8_hh", "k_grad_f16_k128_hh", "k_grad_f16_k128", "k_grad_small")}[mode]
            generic_bf16 = k == "k_grad_bf16" and not (self.K == 64 and self.M % 128 == 0 and self.N % 256 == 0)
            if k not in fast or generic_bf16 or (mode in ("f16x2", "f16x2r") and k == "k_grad_bf16"):
                _notice((mode, k, self.K, self.M % 128 == 0, self.N % 256 == 0),
                        "proxmin_amd: mode %s at %d x %d x %d runs %s, not the tuned kernel (those take K = 64 with M %% 128 = 0 and "
                        "N %% 256 = 0, or K = 128 with M %% 128 = 0 and N %% 128 = 0)" % (mode, self.M, self.N, self.K, k))

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            try:
                if self.mode in ("f16x2", "f16x2r") and self.k1_info()["range_faults"]:
                    _notice(("range", self.M, self.N, self.K),
                            "proxmin_amd: mode %s at %d x %d x %d: the factors ran away from the data (K max|A| max|S| > 2^16 max|Y|: one fp16 "
                            "scale cannot carry that residual); the run went on with the exact-fp32 kernel" % (self.mode, self.M, self.N, self.K))
            except Exception:
                pass
            self.lib.pmx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        _lib.check(self.lib.pmx_ctx_sync(self.h))

    # -- data -------------------------------------------------------------------------------
    def set_Y(self, Y):
        ref = as_device_array(Y)
        if ref is not None:               # already in HBM: float32 adopted in place; float64 copied into the fp64 context's own (padded) array
            assert ref.shape == (self.M, self.N), "Y must be M x N"
            if ref.dtype == np.float64:
                if not self.f64:
                    raise NotImplementedError("a float64 device-resident Y is taken by the fp64 kernels only (library operators and step rules, K <= 128); "
                                              "this call computes in float32: hand it a float32 array")
                _lib.check(self.lib.pmx_set_Y_device_f64(self.h, C.c_void_p(int(ref.ptr)), int(ref.ld)))
                return
            if self.f64:
                raise NotImplementedError("an fp64 context takes a float64 Y")
            self.set_Y_device(ref.ptr, ld=ref.ld, copy=False, keepalive=ref.owner)
            return
        Y = np.asarray(Y)
        assert Y.shape == (self.M, self.N), "Y must be M x N"
        if self.f64:
            Yd = np.ascontiguousarray(Y, dtype=np.float64)
            _lib.check(self.lib.pmx_set_Y_host_f64(self.h, _vp(Yd), self.N))
            return
        Yf = _f32(Y)
        _lib.check(self.lib.pmx_set_Y_host(self.h, _vp(Yf), self.N))

    def set_Y_device(self, dptr, ld=None, copy=False, keepalive=None):
        """Adopt (or copy) a float32 row-major device array, e.g. a torch tensor's data_ptr()."""
        if keepalive is not None:
            self._keep.append(keepalive)
        _lib.check(self.lib.pmx_set_Y_device(self.h, C.c_void_p(int(dptr)), int(ld or self.N), int(bool(copy))))
        if not copy and not self.f64:
            fr = self.k1_info()["frame"]
            if tuple(fr) != (self.M, self.N):
                _notice(("frame-copy", self.M, self.N, self.K),
                        "proxmin_amd: %d x %d x %d runs on the zero-padded frame %d x %d of its tuned kernel: the device array handed over with copy=False is "
                        "COPIED into a frame-sized buffer (%.2f GiB more device memory; PMX_FRAME=0 keeps the caller's array and the guarded kernels)"
                        % (self.M, self.N, self.K, fr[0], fr[1], fr[0] * fr[1] * 4 / 2.0 ** 30))

    def set_W(self, W):
        """Weights of the likelihood (nmf.py:13-41), an M x N array; None goes back to W == 1.  Mode "f32" takes any
        shape; the split-bf16 mode only those of its default kernel (K = 64, M % 128 = 0, N % 256 = 0) and raises
        NotImplementedError otherwise -- see open_weighted()."""
        if W is None:
            _lib.check((self.lib.pmx_set_W_host_f64 if self.f64 else self.lib.pmx_set_W_host)(self.h, None, 0))
            return
        W = np.asarray(W)
        assert W.shape == (self.M, self.N), "W must be M x N"
        if self.f64:                     # [r6] the fp64 matrix-core kernels take weights (mode "f64mfma" forces them on a small problem)
            Wd = np.ascontiguousarray(W, dtype=np.float64)
            _lib.check(self.lib.pmx_set_W_host_f64(self.h, _vp(Wd), self.N))
            return
        Wf = _f32(W)
        _lib.check(self.lib.pmx_set_W_host(self.h, _vp(Wf), self.N))

    def set_W_device(self, dptr, ld=None, copy=False, keepalive=None):
        if keepalive is not None:
            self._keep.append(keepalive)
        _lib.check(self.lib.pmx_set_W_device(self.h, C.c_void_p(int(dptr)), int(ld or self.N), int(bool(copy))))

    def _upload(self, buf, arr2d):
        if self.f64:
            a = np.ascontiguousarray(arr2d, dtype=np.float64)
            _lib.check(self.lib.pmx_upload_f64(self.h, buf, _vp(a), a.size))
            return
        a = _f32(arr2d)
        _lib.check(self.lib.pmx_upload(self.h, buf, _vp(a), a.size))

    def _download(self, buf, rows):
        if self.f64:
            out = np.empty((rows, self.K), dtype=np.float64)
            _lib.check(self.lib.pmx_download_f64(self.h, buf, _vp(out), out.size))
            return out
        out = np.empty((rows, self.K), dtype=np.float32)
        _lib.check(self.lib.pmx_download(self.h, buf, _vp(out), out.size))
        return out

    def set_factors(self, A, S):
        assert A.shape == (self.M, self.K) and S.shape == (self.K, self.N), "A must be M x K and S K x N"
        self._upload(_lib.BUF_A, A)
        self._upload(_lib.BUF_ST, np.asarray(S).T)

    def get_factors(self):
        return self._download(_lib.BUF_A, self.M), self._download(_lib.BUF_ST, self.N).T

    def put(self, base, j, arr):
        """upload block j (0: M x K array, 1: K x N array) of a two-block state (moments, Z/U, ...)."""
        self._upload(base + j, arr if j == 0 else np.asarray(arr).T)

    def get(self, base, j):
        out = self._download(base + j, self.M if j == 0 else self.N)
        return out if j == 0 else out.T

    # -- measurement --------------------------------------------------------------------------
    def set_timing(self, on=True, every=1):
        """Bracket K1 launches with HIP events on the launch stream (every `every`-th launch)."""
        _lib.check(self.lib.pmx_set_timing(self.h, int(every) if on else 0))

    def get_timing(self):
        """(summed K1 duration in ms, number of bracketed K1 launches) since set_timing(True)."""
        ms, n = C.c_double(), C.c_int()
        _lib.check(self.lib.pmx_get_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    PHASES = ("k1_with_prologue", "pack", "collective", "post", "update", "after_update_to_next_k1")

    def set_phase_timing(self, every=4):
        """Row-sharded runs: HIP-event timeline of every `every`-th iteration's phases (include/pmx.h: pmx_set_phase_timing); 0: off."""
        _lib.check(self.lib.pmx_set_phase_timing(self.h, int(every)))

    def get_phase_timing(self):
        """({phase: mean ms}, iterations averaged) since set_phase_timing."""
        ms, n = (C.c_double * 6)(), C.c_int()
        _lib.check(self.lib.pmx_get_phase_timing(self.h, ms, C.byref(n)))
        return {k: ms[i] for i, k in enumerate(self.PHASES)}, n.value

    def time_grad(self, do_A=True, do_S=True, reps=10):
        """average K1 duration in ms (kernel ablation helper, includes the presplit kernel in bf16x3 mode)"""
        ms = C.c_double()
        _lib.check(self.lib.pmx_time_grad(self.h, int(do_A), int(do_S), int(reps), C.byref(ms)))
        return ms.value

    def k1_info(self):
        """Layout of this context's fused residual-gradient kernel (include/pmx.h: pmx_k1_info)."""
        v = (C.c_int * 8)()
        _lib.check(self.lib.pmx_k1_info(self.h, v))
        keys = ("kernel", "chain", "slabs_A", "slabs_S", "row_regions", "col_regions", "panels_per_region", "chain_faults")
        d = dict(zip(keys, list(v)))
        d["kernel"] = ("k_grad_f32", "k_grad_bf16", "k_grad_f16_v8", "k_grad_f16_v9", "k_grad_small", "k_grad_f16_k128", "k_grad_f32_pc", "k64_front", "k_grad_f16_k32", "k_grad_f16_v8_r3", "k_grad_f16_k32_r3", "k_grad_f16_v8_hh", "k_grad_f16_k128_hh", "k64_grad_pass")[d["kernel"]]
        v7 = d.pop("chain_faults")
        d["chain_faults"], d["tail_faults"], d["tail_fused"] = v7 % 1000, (v7 // 1000) % 1000, bool((v7 // 1000000) % 10)
        d["range_faults"] = v7 // 10000000   # 1: a two-term fp16 K1 refused the residual's range, the context went on in exact fp32 (f16_range_fault)
        fr = (C.c_int64 * 3)()
        _lib.check(self.lib.pmx_k1_frame(self.h, fr))
        d["frame"] = (fr[0], fr[1])          # != (M, N): a ragged shape on a zero-padded frame (include/pmx.h: pmx_k1_frame)
        d["frame_K"] = fr[2]                 # != K: K1 runs the next tuned K on zero-padded copies of the factors
        return d

    # -- single operations --------------------------------------------------------------------
    def grad(self):
        """nmf.grad_likelihood at the current device factors -> (gA (M x K), gS (K x N))."""
        _lib.check(self.lib.pmx_grad(self.h))
        return self._download(_lib.BUF_GA, self.M), self._download(_lib.BUF_GST, self.N).T

    def loglike(self):
        out = C.c_double()
        _lib.check(self.lib.pmx_loglike(self.h, C.byref(out)))
        return out.value

    def step_pgm(self):
        out = (C.c_double * 2)()
        _lib.check(self.lib.pmx_step_pgm(self.h, out))
        return out[0], out[1]

    def step_adaprox(self):
        out = np.empty(2 * self.K, dtype=np.float32)
        _lib.check(self.lib.pmx_step_adaprox(self.h, _vp(out)))
        return out[: self.K].copy(), out[self.K:].copy()

    # -- solvers ------------------------------------------------------------------------------
    def pgm_begin(self, prox, accelerated=False, step_scale=1.0, fixed_steps=None, e_rel=(1e-6, 1e-6), bb=None, backtracking=False,
                  host_prox=(False, False), unweighted_rule=False):
        p = _lib.PgmParams()
        p.prox[0], p.prox[1] = prox
        p.accelerated = int(bool(accelerated))
        p.step_scale = float(step_scale)
        p.use_fixed_steps = int(fixed_steps is not None)
        p.unweighted_rule = int(bool(unweighted_rule))    # step_pgm(*X) without its W on a weighted problem (include/pmx.h)
        if fixed_steps is not None:
            p.fixed_steps[0], p.fixed_steps[1] = float(fixed_steps[0]), float(fixed_steps[1])
        p.e_rel[0], p.e_rel[1] = float(e_rel[0]), float(e_rel[1])
        if bb is not None:
            p.bb_type, p.bb_init_r = int(bb[0]), float(bb[1])
        p.backtracking = int(bool(backtracking))
        p.host_prox[0], p.host_prox[1] = int(bool(host_prox[0])), int(bool(host_prox[1]))
        _lib.check(self.lib.pmx_pgm_begin(self.h, C.byref(p)))

    def pgm_split(self, phase, steps=None):
        """One piece of ONE iteration (include/pmx.h: pmx_pgm_split): 0 gradient, 1 argument of the user prox, 2 update."""
        r = _lib.Result()
        st = (C.c_double * 2)(float(steps[0]), float(steps[1])) if steps is not None else None
        _lib.check(self.lib.pmx_pgm_split(self.h, int(phase), st, C.byref(r)))
        return r

    def pgm_bt_split(self, phase):
        """One piece of ONE pgm iteration with the line search when a block's prox is a user callable (include/pmx.h:
        pmx_pgm_bt_split) -> (mask of the blocks whose prox is owed, (T_A s_A, T_S s_S), result)."""
        r = _lib.Result()
        need, eff = C.c_int(), (C.c_double * 2)()
        _lib.check(self.lib.pmx_pgm_bt_split(self.h, int(phase), C.byref(need), eff, C.byref(r)))
        return need.value, (eff[0], eff[1]), r

    def pgm_set_fixed_steps(self, steps):
        """new step constants of a context begun with fixed steps (include/pmx.h: pmx_pgm_set_fixed_steps)"""
        _lib.check(self.lib.pmx_pgm_set_fixed_steps(self.h, (C.c_double * 2)(float(steps[0]), float(steps[1]))))

    def pgm_step_arrays(self, arrays):
        """arrays[j]: None (block j keeps its scalar step) or an array that broadcasts against block j (A: M x K, S: K x N) --
        uploaded element by element for the split phases that follow (include/pmx.h: pmx_pgm_step_arrays)."""
        mask = 0
        for j, a in enumerate(arrays):
            if a is None:
                continue
            shape = (self.M, self.K) if j == 0 else (self.K, self.N)
            self.put(_lib.BUF_STEP_A, j, np.broadcast_to(np.asarray(a, dtype=np.float32), shape))
            mask |= 1 << j
        _lib.check(self.lib.pmx_pgm_step_arrays(self.h, mask))

    def pgm_run(self, n_iter):
        r = _lib.Result()
        _lib.check(self.lib.pmx_pgm_run(self.h, int(n_iter), C.byref(r)))
        return r

    def adaprox_begin(self, prox, scheme="adam", b2=0.999, eps=1e-8, p=0.25, check_convergence=True,
                      prox_max_iter=1000, warm_moments=False, warm_vhat=False, fixed_alpha=None, e_rel=(1e-6, 1e-6),
                      host_step=False, host_prox=(False, False)):
        q = _lib.AdaproxParams()
        q.prox[0], q.prox[1] = prox
        q.scheme = _lib.SCHEME[scheme]
        q.b2, q.eps, q.p = float(b2), float(eps), float(p)
        q.check_convergence = int(bool(check_convergence))
        q.prox_max_iter = int(prox_max_iter)
        q.warm_vhat = int(bool(warm_vhat))
        q.use_fixed_steps = 2 if host_step else int(fixed_alpha is not None)
        q.host_prox[0], q.host_prox[1] = int(bool(host_prox[0])), int(bool(host_prox[1]))
        if fixed_alpha is not None:
            q.fixed_alpha[0], q.fixed_alpha[1] = float(fixed_alpha[0]), float(fixed_alpha[1])
        q.e_rel[0], q.e_rel[1] = float(e_rel[0]), float(e_rel[1])
        _lib.check(self.lib.pmx_adaprox_begin(self.h, C.byref(q), int(bool(warm_moments))))

    def adaprox_run(self, b1_slice, b1_prev):
        b1 = np.ascontiguousarray(b1_slice, dtype=np.float64)
        r = _lib.Result()
        _lib.check(self.lib.pmx_adaprox_run(self.h, int(b1.size), b1.ctypes.data_as(C.POINTER(C.c_double)),
                                            float(b1_prev), C.byref(r)))
        return r

    def adaprox_set_alpha(self, alpha_A, alpha_S):
        """per-component step sizes of the next iteration (a user `step` callable's return value, K values per block)"""
        a = np.ascontiguousarray(np.concatenate([np.asarray(alpha_A, np.float32).ravel(), np.asarray(alpha_S, np.float32).ravel()]))
        assert a.size == 2 * self.K
        _lib.check(self.lib.pmx_adaprox_set_alpha(self.h, _vp(a)))

    def adaprox_split(self, phase, it, b1_it, b1_prev, host_tau=(0, 0)):
        """One half of ONE iteration (include/pmx.h: pmx_adaprox_split).  Returns (Result, (maxpsi_A, maxpsi_S))."""
        r = _lib.Result()
        tau = (C.c_int * 2)(int(host_tau[0]), int(host_tau[1]))
        mp = (C.c_double * 2)()
        _lib.check(self.lib.pmx_adaprox_split(self.h, int(phase), int(it), float(b1_it), float(b1_prev), tau, mp, C.byref(r)))
        return r, (mp[0], mp[1])

    def bsdmm_begin(self, prox_f, proxs_g, e_rel=(1e-6, 1e-6), e_abs=(0.0, 0.0), update_order=None):
        p = _lib.BsdmmParams()
        if update_order is not None:
            order = [int(j) for j in update_order]
            if len(order) > 8:
                raise NotImplementedError("update_order with more than 8 entries")
            p.n_order = len(order)
            for i, j in enumerate(order):
                p.order[i] = j
        p.prox_f[0], p.prox_f[1] = prox_f
        for j in range(2):
            g = proxs_g[j] or []
            if len(g) > _lib.MAX_G:
                raise NotImplementedError("at most %d constraints per factor on the device" % _lib.MAX_G)
            p.n_g[j] = len(g)
            for i, ps in enumerate(g):
                p.prox_g[j][i] = ps
            p.e_rel[j], p.e_abs[j] = float(e_rel[j]), float(e_abs[j])
        _lib.check(self.lib.pmx_bsdmm_begin(self.h, C.byref(p)))

    def set_host_grad(self, on=True):
        """The gradient of every iteration is what the caller uploads into BUF_GA / BUF_GST (a user `grad` callable);
        the context needs no Y (include/pmx.h: pmx_set_host_grad)."""
        _lib.check(self.lib.pmx_set_host_grad(self.h, int(bool(on))))

    def bsdmm_split(self, j, phase, host_f=False, host_g=0, last_block=False, step_f=0.0):
        """One piece of one block update (include/pmx.h: pmx_bsdmm_split)."""
        r = _lib.Result()
        _lib.check(self.lib.pmx_bsdmm_split(self.h, int(j), int(phase), int(bool(host_f)), int(host_g), int(bool(last_block)), float(step_f), C.byref(r)))
        return r

    def bsdmm_run(self, n_iter):
        r = _lib.Result()
        _lib.check(self.lib.pmx_bsdmm_run(self.h, int(n_iter), C.byref(r)))
        return r


def open_weighted(M, N, K, W, **kw):
    """Context for a weighted likelihood: the default arithmetic mode where its kernel takes weights, exact fp32
    otherwise (never a CPU path: both are device kernels)."""
    dev = DeviceNMF(M, N, K, **kw)
    if W is None:
        return dev
    try:
        dev.set_W(W)
        return dev
    except NotImplementedError:
        mode = dev.mode
        dev.close()
    _notice(("weighted", mode, K), "proxmin_amd: a weighted likelihood at %d x %d x %d has no kernel in mode %s (its weighted kernels take "
            "K = 64, M %% 128 = 0, N %% 256 = 0): this context computes in exact fp32 (k_grad_f32)" % (M, N, K, mode))
    dev = DeviceNMF(M, N, K, **dict(kw, mode="f32"))
    dev.set_W(W)
    return dev
