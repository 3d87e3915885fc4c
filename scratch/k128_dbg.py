import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch, bench
from proxmin_amd import engine
K = 128
M, N = 8192, 16384
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
    os.environ["PMX_K128_DBG"] = str(dbg)
    with engine.DeviceNMF(M, N, K, mode="f16x2") as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        print("dbg=%d (1 noY 2 noGEMM 4 noEPI)  resid-only %.4f ms   full %.4f ms" % (dbg, dev.time_grad(do_A=0, do_S=0, reps=20), dev.time_grad(do_A=1, do_S=1, reps=20)), flush=True)
