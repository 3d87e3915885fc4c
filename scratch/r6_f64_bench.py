"""[r6] fp64 at size (k_big_f64.hip): ms per iteration of the three back-ends on fp64 inputs; run under rocprofv3 --kernel-trace --stats for the kernels' shares"""
import sys, os, time
from functools import partial
sys.path.insert(0, os.getcwd())
import numpy as np
import proxmin_amd as pm
from proxmin_amd.engine import DeviceNMF
ops = pm.operators
cfgs = {"cfg3": (16384, 16384, 64), "cfg2": (4096, 4096, 32), "cfg4share": (8192, 16384, 128), "cfg5": (8192, 8192, 64), "mid": (4096, 8192, 64)}
which = sys.argv[1:] or ["cfg2", "cfg3"]
rng = np.random.default_rng(1)
for name in which:
    M, N, K = cfgs[name]
    A = rng.random((M, K)); S = rng.random((K, N)); S /= S.sum(0)
    Y = A @ S + 0.01 * rng.standard_normal((M, N))
    A0 = rng.random((M, K)); S0 = rng.random((K, N)); S0 /= S0.sum(0)
    with DeviceNMF(M, N, K, mode="f64") as dev:
        t = time.time(); dev.set_Y(Y); dev.set_factors(A0, S0); up = time.time() - t
        info = dev.k1_info()
        # pgm
        dev.pgm_begin([ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(ops.prox_plus, 1)], e_rel=(1e-12, 1e-12))
        dev.pgm_run(2)
        t = time.time(); dev.pgm_run(8); pgm = (time.time() - t) / 8
        dev.set_factors(A0, S0)
        dev.adaprox_begin([ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_unity_plus, axis=0), 1)], scheme="amsgrad", check_convergence=False, e_rel=(1e-3, 1e-3))
        dev.adaprox_run(np.full(2, 0.9), 0.9)
        t = time.time(); r = dev.adaprox_run(np.full(8, 0.9), 0.9); ada = (time.time() - t) / 8
        dev.set_factors(A0, S0)
        dev.bsdmm_begin([ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(ops.prox_plus, 1)], [[ops.device_proxseq(ops.prox_plus, 0)], [ops.device_proxseq(ops.prox_plus, 1)]], e_rel=(1e-12, 1e-12), e_abs=(0.0, 0.0))
        dev.bsdmm_run(2)
        t = time.time(); dev.bsdmm_run(8); bsd = (time.time() - t) / 8
    fl = 8.0 * M * N * K
    print("%s %dx%dx%d slabs %d/%d upload %.2fs | pgm %.3f ms/it | adaprox %.3f ms/it (sub %s) | bsdmm %.3f ms/it | two-pass K1 at 78.6 TF: %.3f ms" % (
        name, M, N, K, info["slabs_A"], info["slabs_S"], up, pgm * 1e3, ada * 1e3, list(r.sub_iterations), bsd * 1e3, fl / 78.6e12 * 1e3), flush=True)
