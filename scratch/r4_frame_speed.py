"""Ragged shapes: K1 and a whole iteration on the zero-padded frame against the guarded kernels (PMX_FRAME=0).  Device-resident Y,
solver loop timed on the device side (bench.py's helpers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import numpy as np, torch
import bench
from proxmin_amd import engine

def run(M, N, K, mode, algo, unity):
    Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
    with engine.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        info = dev.k1_info()
        k1 = dev.time_grad(do_A=1, do_S=1, reps=20)
        step = bench.begin_solver(dev, algo, unity)
        step(10)
        dev.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(50)
        dev.sync()
        it = (time.perf_counter() - t0) / 50 * 1e3
    return info, k1, it

CASES = ((16000, 16000, 64, "f16x2", "adaprox", True), (16000, 16000, 64, "f32", "adaprox", True), (5000, 7000, 64, "f16x2", "adaprox", True),
         (5000, 7000, 64, "f32", "pgm", False), (10000, 12000, 128, "f16x2", "adaprox", True), (4000, 4000, 32, "f16x2", "pgm", False),
         (4000, 4000, 32, "f32", "pgm", False), (16000, 16000, 64, "f16x2", "bsdmm", False))
if len(sys.argv) > 1 and sys.argv[1] == "K":        # K between the tuned ones (the frame pads the components as well)
    CASES = ((16384, 16384, 50, "f16x2", "adaprox", True), (16384, 16384, 50, "f32", "adaprox", True), (16384, 16384, 100, "f16x2", "adaprox", True),
             (16000, 16000, 40, "f16x2", "adaprox", True), (4096, 4096, 20, "f16x2", "pgm", False), (4096, 4096, 20, "f32", "pgm", False),
             (16384, 16384, 50, "f16x2", "bsdmm", False), (8192, 8192, 10, "f16x2", "pgm", False))
for M, N, K, mode, algo, unity in CASES:
    for fr in ("1", "0"):
        os.environ["PMX_FRAME"] = fr
        info, k1, it = run(M, N, K, mode, algo, unity)
        print("%6d x %6d x %3d %-6s %-7s frame=%s %-16s %-14s K %3d chain %2d slabs %3d/%d grid %dx%d  K1 %.4f ms  iteration %.4f ms" % (
            M, N, K, mode, algo, fr, info["kernel"], info["frame"], info["frame_K"], info["chain"], info["slabs_A"], info["slabs_S"], info["row_regions"], info["col_regions"], k1, it), flush=True)
