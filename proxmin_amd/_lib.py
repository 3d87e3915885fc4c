"""ctypes binding of libpmx.so (the C ABI declared in include/pmx.h).

The shared library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950) and
lives next to this file.  There is NO CPU fallback: if the library is missing, or no MI355X is
visible, the package fails loudly at the first call that needs the device.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMX_LIB") or os.path.join(_HERE, "libpmx.so")   # PMX_LIB: A/B a second build (tuning)

MAX_SEQ = 4
MAX_G = 4
MAXK = 128
ABI_VERSION = 5

# enums (include/pmx.h)
MODE_F32, MODE_BF16X3, MODE_F16X2, MODE_F64, MODE_F16X2R, MODE_F64_MFMA = 0, 2, 3, 4, 5, 6
PROX = {"id": 0, "zero": 1, "plus": 2, "unity": 3, "unity_plus": 4, "min": 5, "max": 6,
        "hard": 7, "hard_plus": 8, "soft": 9, "soft_plus": 10}
SCHEME = {"adam": 0, "nadam": 1, "amsgrad": 2, "padam": 3, "adamx": 4, "radam": 5}
BUF_A, BUF_ST, BUF_GA, BUF_GST, BUF_MA, BUF_MST, BUF_VA, BUF_VST, BUF_VHA, BUF_VHST = range(10)
BUF_EVAL_A, BUF_EVAL_ST, BUF_TMP_A, BUF_TMP_ST, BUF_PSI_A, BUF_PSI_ST = 10, 11, 12, 13, 14, 15
BUF_Z0, BUF_U0, BUF_TG0 = 16, 32, 48
BUF_BT_A = 66                         # + block: argument / result of a user prox inside the line search (pmx_pgm_bt_split)
BUF_STEP_A = 64                       # + block: per-element steps of a user `step` that returned arrays (pgm)


class Prox(C.Structure):
    _fields_ = [("op", C.c_int32), ("unit", C.c_int32), ("thresh", C.c_double), ("relative", C.c_int32), ("reserved", C.c_int32)]


class ProxSeq(C.Structure):
    _fields_ = [("n", C.c_int32), ("repeat", C.c_int32), ("seq", Prox * MAX_SEQ)]


class PgmParams(C.Structure):
    _fields_ = [("prox", ProxSeq * 2), ("accelerated", C.c_int32), ("step_scale", C.c_float),
                ("use_fixed_steps", C.c_int32), ("unweighted_rule", C.c_int32), ("fixed_steps", C.c_double * 2), ("e_rel", C.c_double * 2),
                ("bb_type", C.c_int32), ("bb_init_r", C.c_double), ("backtracking", C.c_int32), ("host_prox", C.c_int32 * 2)]


class Result(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("total_iterations", C.c_int32), ("stopped", C.c_int32),
                ("converged", C.c_int32 * 2), ("steps", C.c_double * 2), ("sub_iterations", C.c_int64 * 2)]


class AdaproxParams(C.Structure):
    _fields_ = [("prox", ProxSeq * 2), ("scheme", C.c_int32), ("b2", C.c_double), ("eps", C.c_double),
                ("p", C.c_double), ("check_convergence", C.c_int32), ("prox_max_iter", C.c_int32),
                ("warm_vhat", C.c_int32), ("use_fixed_steps", C.c_int32), ("fixed_alpha", C.c_double * 2),
                ("e_rel", C.c_double * 2), ("host_prox", C.c_int32 * 2)]


class BsdmmParams(C.Structure):
    _fields_ = [("prox_f", ProxSeq * 2), ("n_g", C.c_int32 * 2), ("prox_g", (ProxSeq * MAX_G) * 2),
                ("e_rel", C.c_double * 2), ("e_abs", C.c_double * 2), ("n_order", C.c_int32), ("order", C.c_int32 * 8)]


_SIGNATURES = {
    "pmx_abi_version": (C.c_int, []),
    "pmx_last_error": (C.c_char_p, []),
    "pmx_device_count": (C.c_int, []),
    "pmx_abi_sizes": (C.c_int, [C.POINTER(C.c_int)]),
    "pmx_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "pmx_ctx_destroy": (C.c_int, [C.c_void_p]),
    "pmx_ctx_sync": (C.c_int, [C.c_void_p]),
    "pmx_set_Y_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_set_Y_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "pmx_set_W_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_set_W_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "pmx_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "pmx_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "pmx_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pmx_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pmx_set_Y_host_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_set_Y_device_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_set_W_host_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_upload_f64": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "pmx_download_f64": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "pmx_set_phase_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pmx_get_phase_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pmx_time_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pmx_k1_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pmx_k1_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "pmx_grad": (C.c_int, [C.c_void_p]),
    "pmx_loglike": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pmx_step_pgm": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pmx_step_adaprox": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pmx_prox_apply": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(ProxSeq), C.c_void_p]),
    "pmx_prox_array": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.POINTER(ProxSeq), C.c_void_p]),
    "pmx_bb_sums": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]),
    "pmx_pgm_begin": (C.c_int, [C.c_void_p, C.POINTER(PgmParams)]),
    "pmx_pgm_run": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Result)]),
    "pmx_pgm_split": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(Result)]),
    "pmx_pgm_bt_split": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(Result)]),
    "pmx_adaprox_set_alpha": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pmx_adaprox_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(Result)]),
    "pmx_adaprox_begin": (C.c_int, [C.c_void_p, C.POINTER(AdaproxParams), C.c_int]),
    "pmx_adaprox_run": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.POINTER(Result)]),
    "pmx_bsdmm_begin": (C.c_int, [C.c_void_p, C.POINTER(BsdmmParams)]),
    "pmx_bsdmm_run": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Result)]),
    "pmx_set_world": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64]),
    "pmx_comm_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pmx_set_comm_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_adaprox_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "pmx_pgm_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pmx_bsdmm_phase": (C.c_int, [C.c_void_p, C.c_int]),
    "pmx_chain_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pmx_adaprox_more_subs": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pmx_iter_result": (C.c_int, [C.c_void_p, C.POINTER(Result)]),
    "pmx_set_host_grad": (C.c_int, [C.c_void_p, C.c_int]),
    "pmx_set_s_split": (C.c_int, [C.c_void_p, C.c_int]),
    "pmx_comm_layout_split": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pmx_set_comm_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_buffer_ptr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "pmx_bsdmm_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_double, C.POINTER(Result)]),
    "pmx_pgm_step_arrays": (C.c_int, [C.c_void_p, C.c_int]),
    "pmx_pgm_set_fixed_steps": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pmx_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pmx_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "pmx_comm_all_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_comm_reduce_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "pmx_comm_destroy": (C.c_int, [C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class PmxError(RuntimeError):
    pass


def load():
    """Load libpmx.so (once) and attach the signatures.  Raises if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PmxError("libpmx.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). proxmin_amd has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64: a process that imports torch AFTER this library
    # (which resolves the system runtime) ends up with two runtimes, and torch then sees no device; with torch imported FIRST
    # libpmx binds to the copy already in the process.  This package needs NumPy and libamdhip64 only and does NOT import torch
    # on its own account [r5: the preload used to be the default].  A process that uses both imports torch first -- or sets
    # PMX_TORCH_PRELOAD=1 and lets this loader do it (tests/conftest.py and bench.py do; proxmin_amd.distributed, which needs
    # torch.distributed, goes through require_torch() below and says what is wrong instead of failing inside torch).
    if "torch" not in sys.modules and os.environ.get("PMX_TORCH_PRELOAD", "0") == "1":
        import torch  # noqa: F401
    elif "torch" not in sys.modules and os.environ.get("PMX_TORCH_PRELOAD") is None:
        # [r6] the silent failure mode: this process loads libpmx now and may import torch LATER -- torch would then see no GPU
        # and say nothing.  If torch is installed, say so once, at the moment the order is decided (find_spec does not import it).
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import logging
                logging.getLogger("proxmin").info(
                    "proxmin_amd: libpmx.so is being loaded before torch.  If this process imports torch LATER, torch will see no GPU (PyTorch-ROCm "
                    "brings its own HIP runtime): import torch first, or set PMX_TORCH_PRELOAD=1 (PMX_TORCH_PRELOAD=0 silences this note)")
                _install_late_torch_guard()
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pmx_abi_version() != ABI_VERSION:
        raise PmxError("libpmx.so ABI %d != expected %d; rebuild" % (lib.pmx_abi_version(), ABI_VERSION))
    sizes = (C.c_int * 5)()
    lib.pmx_abi_sizes(sizes)
    mine = [C.sizeof(t) for t in (ProxSeq, PgmParams, AdaproxParams, BsdmmParams, Result)]
    if list(sizes) != mine:
        raise PmxError("struct layouts of libpmx.so %r differ from the ctypes mirrors %r; rebuild" % (list(sizes), mine))
    _lib = lib
    return lib


class _LateTorchGuard:
    """meta-path finder that imports nothing: it notices `torch` being looked up after libpmx.so was loaded without it on a box with a GPU and says
    what will go wrong -- once, as a WARNING on logger "proxmin", at the import instead of at the first tensor that silently lands on the CPU.
    (It does not raise: libraries probe for torch with importlib.util.find_spec, and a probe is not an import.)"""
    warned = False

    def find_spec(self, name, path=None, target=None):
        if name == "torch" and not _LateTorchGuard.warned and _lib is not None and "torch" not in sys.modules:
            try:
                has_gpu = _lib.pmx_device_count() >= 1
            except Exception:
                has_gpu = False
            if has_gpu:
                _LateTorchGuard.warned = True
                import logging
                logging.getLogger("proxmin").warning(
                    "proxmin_amd: torch is being imported AFTER libpmx.so was loaded: PyTorch-ROCm brings its own HIP runtime and, loaded second, sees no "
                    "GPU.  Import torch before the first proxmin_amd call, or set PMX_TORCH_PRELOAD=1")
        return None


def _install_late_torch_guard():
    if not any(isinstance(f, _LateTorchGuard) for f in sys.meta_path):
        sys.meta_path.insert(0, _LateTorchGuard())


def require_torch():
    """torch for the callers that need it (proxmin_amd.distributed, device-array helpers): imported here if the library is not
    loaded yet, otherwise it must be in the process already (see load(): one HIP runtime per process)."""
    if "torch" not in sys.modules and _lib is not None:
        raise PmxError("torch must be imported BEFORE the first proxmin_amd call that loads libpmx.so (PyTorch-ROCm brings its own "
                       "HIP runtime; loaded second it sees no device): `import torch` first, or set PMX_TORCH_PRELOAD=1")
    import torch
    return torch


def check(rc):
    if rc != 0:
        msg = load().pmx_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise AssertionError(msg)          # the reference validates arguments with `assert`
        if rc == -4:
            raise NotImplementedError(msg)
        raise PmxError("libpmx error %d: %s" % (rc, msg))


def require_gpu():
    lib = load()
    if lib.pmx_device_count() < 1:
        raise PmxError("no HIP device visible: proxmin_amd runs only on an MI355X (gfx950); there is no CPU path")
    return lib
