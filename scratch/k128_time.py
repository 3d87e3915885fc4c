"""k_grad_f16_k128: launch time with its gradient halves switched off one at a time (8192-row share of cfg4, and all of it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch, bench
from proxmin_amd import engine
K = 128
for M, N in ((8192, 16384), (65536, 16384)):
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    with engine.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        print(M, N, dev.k1_info(), flush=True)
        for dA, dS in ((0, 0), (0, 1), (1, 0), (1, 1)):
            print("   doA=%d doS=%d %.4f ms" % (dA, dS, dev.time_grad(do_A=dA, do_S=dS, reps=20)), flush=True)
    del Y
