# scratch: phase cycle sums of K1 (PMX_K1_PROF=1: wave 0 = a producer, wave 4 = a consumer; k_grad_f16_v8's PH marks) at cfg3, modes f16x2r <HH> and f16x2
import sys, os
os.environ["PMX_K1_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd import engine as eng, _lib
M = N = 16384; K = 64
rng = np.random.default_rng(1)
A = rng.random((M, K), dtype=np.float32); S = rng.random((K, N), dtype=np.float32); S /= S.sum(0, keepdims=True)
Y = (A @ S + 0.01 * rng.standard_normal((M, N)).astype(np.float32)).astype(np.float32)
for name, mode, env in (("HH chained", "f16x2r", {}), ("HH slabs", "f16x2r", {"PMX_K1_CHAIN": "0"}), ("f16x2 slabs", "f16x2", {"PMX_K1_CHAIN": "0"})):
    os.environ.update(env)
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y(Y); dev.set_factors(A, S)
        ms = C.c_double()
        sys.stderr.write("== %s (%s)\n" % (name, dev.k1_info()["kernel"])); sys.stderr.flush()
        _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 50, C.byref(ms)))
        sys.stderr.write("   gradient pass %.4f ms; slots per workgroup: %d\n" % (ms.value, dev.k1_info()["panels_per_region"] * 8)); sys.stderr.flush()
    for k in env: del os.environ[k]
