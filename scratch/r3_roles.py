# scratch (round 3): K1 role ablations of the CURRENT build (doA / doS switch the consumers' two contractions off)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, device=0, mode="f16x2")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
run = bench.begin_solver(dev, backend, unity)
run(10)
print(dev.k1_info())
for rep in range(2):
    for doA, doS in ((1, 1), (0, 0), (1, 0), (0, 1)):
        print("doA=%d doS=%d  %.4f ms" % (doA, doS, dev.time_grad(doA, doS, 300)), flush=True)
