"""k_grad_f32_pc (exact fp32, producer / consumer wavefronts): parity against fp64 NumPy and launch time against k_grad_f32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
import torch
from proxmin_amd import engine
from oracle import nmf_oracle as orc

def check(M, N, K, weighted=False):
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=M + N + K)
    W = None
    if weighted:
        rng = np.random.default_rng(3)
        W = (0.1 + 2 * rng.random((M, N))).astype(np.float32)
        W[rng.random((M, N)) < 0.1] = 0
    with engine.DeviceNMF(M, N, K, mode="f32") as dev:
        info = dev.k1_info()
        dev.set_Y(Y)
        if W is not None:
            dev.set_W(W)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
        loss = dev.loglike()
    x64 = [x.astype(np.float64) for x in (A, S, Y)] + ([W.astype(np.float64)] if weighted else [])
    rA, rS = orc.residual_gradients(*x64)
    rl = orc.half_sq_residual(*x64)
    print("M=%d N=%d K=%d W=%d kernel=%s regions=%dx%d RP=%d  errA=%.2e errS=%.2e errL=%.2e" % (
        M, N, K, weighted, info["kernel"], info["row_regions"], info["col_regions"], info["panels_per_region"],
        np.abs(gA - rA).max() / np.abs(rA).max(), np.abs(gS - rS).max() / np.abs(rS).max(), abs(loss - rl) / rl), flush=True)

for M, N, K in ((128, 256, 64), (1024, 768, 64), (2304, 4096, 64), (128, 256, 32), (5120, 1024, 32), (4096, 4096, 32)):
    check(M, N, K)
check(1024, 768, 64, True)
check(640, 512, 32, True)

import bench
for M, N, K in ((16384, 16384, 64), (4096, 4096, 32), (4096, 4096, 64)):
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    for pc in ("1", "0"):
        os.environ["PMX_K1_F32PC"] = pc
        with engine.DeviceNMF(M, N, K, mode="f32") as dev:
            dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
            dev.set_factors(A0, S0)
            print(M, N, K, dev.k1_info()["kernel"], "%.4f ms" % dev.time_grad(do_A=1, do_S=1, reps=20), flush=True)
    del Y
