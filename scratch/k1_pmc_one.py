# one K1 configuration (doA, doS from argv) launched 12 times back to back: attribute PMC counters to the roles of the kernel
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMX_K1_CHAIN", "0")
import __graft_entry__ as g
g.build()
import torch, bench
from proxmin_amd import engine
dA, dS = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "f16x2"
M = N = 16384
Y, A0, S0 = bench.make_problem_device(M, N, 64, True, 1234, torch.device("cuda", 0))
with engine.DeviceNMF(M, N, 64, mode=mode) as dev:
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    print("doA=%d doS=%d %s %.4f ms" % (dA, dS, dev.k1_info()["kernel"], dev.time_grad(do_A=dA, do_S=dS, reps=12)), flush=True)
