import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMX_K1_PROF"] = "1"
os.environ["PMX_K1_CHAIN"] = "0"
import __graft_entry__ as g
g.build()
import torch, bench
from proxmin_amd import engine
M = N = 16384
Y, A0, S0 = bench.make_problem_device(M, N, 64, True, 1234, torch.device("cuda", 0))
with engine.DeviceNMF(M, N, 64, mode="f16x2") as dev:
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    for dA, dS in ((0, 0), (1, 1)):
        print("doA=%d doS=%d %.4f ms" % (dA, dS, dev.time_grad(do_A=dA, do_S=dS, reps=20)), flush=True)
