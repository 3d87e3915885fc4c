# scratch: K1 time of an ablation build (PMX_LIB=...): cfg3 gradient pass in mode f16x2r (results are garbage by construction: only the time counts)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from proxmin_amd import engine as eng, _lib
M = N = 16384; K = 64
rng = np.random.default_rng(1)
A = rng.random((M, K), dtype=np.float32); S = rng.random((K, N), dtype=np.float32); S /= S.sum(0, keepdims=True)
Y = (A @ S + 0.01 * rng.standard_normal((M, N)).astype(np.float32)).astype(np.float32)
for mode in ("f16x2r", "f16x2"):
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y(Y); dev.set_factors(A, S)
        ms = C.c_double()
        _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 30, C.byref(ms)))
        _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 100, C.byref(ms)))
        print("%s %s: gradient pass %.4f ms [%s]" % (os.environ.get("PMX_LIB", "libpmx.so"), mode, ms.value, dev.k1_info()["kernel"]), flush=True)
