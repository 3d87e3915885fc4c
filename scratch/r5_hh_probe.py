# scratch: mode f16x2r's <HH> K1 (the residual from the high x high product + the K x K correction slab, k_gfix.hip; PMX_F16_R3=2, the default of the mode)
# beside <R3> (PMX_F16_R3=1), plain f16x2 and exact fp32: gradient error against fp64 and K1 time.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd import engine as eng
from oracle import nmf_oracle as orc
SHAPES = ((2048, 2048, 64), (1024, 4096, 64), (1000, 1500, 50), (2048, 2048, 128))
if len(sys.argv) > 1 and sys.argv[1] == "time":
    SHAPES = ()
for M, N, K in SHAPES:
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, unity_S=True, seed=4321)
    r64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    line = "%d x %d x %d  gA / gS rms error of max|g|:" % (M, N, K)
    for name, mode, env in (("f32", "f32", {}), ("f16x2", "f16x2", {}), ("R3", "f16x2r", {"PMX_F16_R3": "1"}), ("HH", "f16x2r", {"PMX_F16_R3": "2"})):
        os.environ.update(env)
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A, S)
            gr = dev.grad(); g2 = dev.grad()
            assert np.array_equal(gr[0], g2[0]) and np.array_equal(gr[1], g2[1])
            info = dev.k1_info()
        for k in env: del os.environ[k]
        e = [float(np.sqrt(((gr[j] - r64[j]) ** 2).mean()) / np.abs(r64[j]).max()) for j in range(2)]
        line += "  %s (%s) %.2e / %.2e |" % (name, str(info["kernel"]).replace("k_grad_", ""), e[0], e[1])
    print(line, flush=True)
# K1 time at cfg3
import ctypes as C
from proxmin_amd import _lib
M = N = 16384; K = 64
rng = np.random.default_rng(1)
A = rng.random((M, K), dtype=np.float32); S = rng.random((K, N), dtype=np.float32); S /= S.sum(0, keepdims=True)
Y = (A @ S + 0.01 * rng.standard_normal((M, N)).astype(np.float32)).astype(np.float32)
A0 = rng.random((M, K), dtype=np.float32); S0 = rng.random((K, N), dtype=np.float32); S0 /= S0.sum(0, keepdims=True)
for rnd in range(2):
    for name, mode, env in (("f16x2", "f16x2", {}), ("R3", "f16x2r", {"PMX_F16_R3": "1"}), ("HH", "f16x2r", {"PMX_F16_R3": "2"})):
        os.environ.update(env)
        with eng.DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A0, S0)
            ms = C.c_double()
            _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 30, C.byref(ms)))
            _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 100, C.byref(ms)))
            print("cfg3 %s: gradient pass (K1 + whatever rides with it) %.4f ms  [%s]" % (name, ms.value, dev.k1_info()["kernel"]), flush=True)
        for k in env: del os.environ[k]
# the correction in the launch stream instead of beside K1
os.environ["PMX_GFIX_SIDE"] = "0"; os.environ["PMX_F16_R3"] = "2"
with eng.DeviceNMF(M, N, K, mode="f16x2r") as dev:
    dev.set_Y(Y); dev.set_factors(A0, S0)
    ms = C.c_double()
    _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 30, C.byref(ms)))
    _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 100, C.byref(ms)))
    print("cfg3 HH, correction in the launch stream: %.4f ms" % ms.value, flush=True)
# cfg4's share: K1 time at K = 128
M, N, K = 8192, 16384, 128
rng = np.random.default_rng(2)
A0 = rng.random((M, K), dtype=np.float32); S0 = rng.random((K, N), dtype=np.float32) * np.float32(16.0 / K)
Y = rng.random((M, N), dtype=np.float32) * 16
for k in ("PMX_GFIX_SIDE", "PMX_F16_R3"):
    os.environ.pop(k, None)
for name, mode in (("f16x2", "f16x2"), ("HH", "f16x2r")) * 2:
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y(Y); dev.set_factors(A0, S0)
        ms = C.c_double()
        _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 30, C.byref(ms)))
        _lib.check(dev.lib.pmx_time_grad(dev.h, 1, 1, 100, C.byref(ms)))
        print("cfg4 share %s: gradient pass %.4f ms  [%s]" % (name, ms.value, dev.k1_info()["kernel"]), flush=True)
