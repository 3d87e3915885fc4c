# [r6] is pgm at medium K = 64 shapes latency-bound like cfg2?  per-iteration time and K1 share, modes f32 / f16x2r
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
for (M, N, K) in ((4096, 4096, 64), (8192, 8192, 64), (4096, 4096, 128), (2048, 4096, 32)):
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    for mode in ("f32", "f16x2r"):
        dev = DeviceNMF(M, N, K, mode=mode)
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, "pgm", False)
        run(40)
        dev.set_timing(True, every=4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(400)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, n = dev.get_timing()
        print("%5d x %5d x %3d %-7s %-20s %.1f us per iteration, K1 %.1f us, rest %.1f us" % (M, N, K, mode, dev.k1_info()["kernel"], 1e6 * dt / 400, 1e3 * ms / max(n, 1), 1e6 * dt / 400 - 1e3 * ms / max(n, 1)), flush=True)
        dev.close()
    del Y
