"""K = 128 two-term fp16 K1 (k_grad_f16_k128): parity against fp64 NumPy and launch time, one pass and two passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
import torch
from proxmin_amd import engine
from oracle import nmf_oracle as orc

def check(M, N, passes):
    K = 128
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=M + N)
    with engine.DeviceNMF(M, N, K, mode="f16x2") as dev:
        info = dev.k1_info()
        dev.set_Y(Y); dev.set_factors(A, S)
        gA, gS = dev.grad()
        loss = dev.loglike()
    A64, S64, Y64 = (x.astype(np.float64) for x in (A, S, Y))
    rA, rS = orc.residual_gradients(A64, S64, Y64)
    eA = np.abs(gA - rA).max() / np.abs(rA).max(); eS = np.abs(gS - rS).max() / np.abs(rS).max()
    el = abs(loss - orc.half_sq_residual(A64, S64, Y64)) / orc.half_sq_residual(A64, S64, Y64)
    print("M=%d N=%d passes=%d kernel=%s regions=%dx%d RP=%d  errA=%.2e errS=%.2e errL=%.2e" % (
        M, N, passes, info["kernel"], info["row_regions"], info["col_regions"], info["panels_per_region"], eA, eS, el), flush=True)

for passes in (1,):
    for M, N in ((128, 128), (512, 512), (2048, 1024), (1024, 4096), (3200, 2176)):
        check(M, N, passes)

import bench
M, N, K = 8192, 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
for mode, passes in (("f32", 1), ("f16x2", 1)):
    with engine.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        print(mode, "passes", passes, dev.k1_info()["kernel"], "%.4f ms" % dev.time_grad(do_A=1, do_S=1, reps=20), flush=True)
