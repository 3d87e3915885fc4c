# quick parity + timing check of k_grad_f16_v9 against the fp64 oracle and v8 (same box)
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from proxmin_amd import engine
from oracle import nmf_oracle as orc

def grads(M, N, v9, chain):
    os.environ["PMX_K1_V9"] = str(v9)
    os.environ["PMX_K1_CHAIN"] = str(chain)
    Y, A, S = orc.synthetic_problem(M, N, 64, np.float32, seed=M + N)
    A[:, 0] += np.linspace(0.0, 1.0, M, dtype=np.float32)
    S[63, :] += np.linspace(1.0, 0.0, N, dtype=np.float32)
    with engine.DeviceNMF(M, N, 64, mode="f16x2") as dev:
        info = dev.k1_info()
        dev.set_Y(Y)
        dev.set_factors(A, S)
        gA, gS = dev.grad()
        loss = dev.loglike()
        ms = dev.time_grad(reps=20)
        info2 = dev.k1_info()
    return Y, A, S, gA, gS, loss, ms, info, info2

for M, N in ((128, 256), (4096, 4096), (4096, 16384), (16384, 16384)):
    for v9, chain in ((1, 0), (1, 32), (0, 32)):
        Y, A, S, gA, gS, loss, ms, info, info2 = grads(M, N, v9, chain)
        if M * N <= 4096 * 16384:
            rA, rS = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
            eA = np.abs(gA - rA).max() / np.abs(rA).max()
            eS = np.abs(gS - rS).max() / np.abs(rS).max()
            el = abs(loss / orc.half_sq_residual(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64)) - 1)
        else:
            eA = eS = el = float("nan")
        print("M=%d N=%d v9=%d chain=%d -> kernel %s chain %d slabs %d/%d faults %d | errA %.2e errS %.2e loss %.1e | K1 %.4f ms" % (
            M, N, v9, chain, info["kernel"], info["chain"], info["slabs_A"], info["slabs_S"], info2["chain_faults"], eA, eS, el, ms), flush=True)
