# where does a block's time go?  K1 with only some of its outputs (pmx_time_grad), v9 against v8, cfg3 shape, same box
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch
import bench
from proxmin_amd import engine
M = N = 16384
Y, A0, S0 = bench.make_problem_device(M, N, 64, True, 1234, torch.device("cuda", 0))
for v9, chain in ((1, 0), (0, 0)):
    os.environ["PMX_K1_V9"] = str(v9)
    os.environ["PMX_K1_CHAIN"] = str(chain)
    with engine.DeviceNMF(M, N, 64, mode="f16x2") as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        info = dev.k1_info()
        out = []
        for dA, dS in ((0, 0), (0, 1), (1, 0), (1, 1)):
            out.append("doA=%d doS=%d %.4f ms" % (dA, dS, dev.time_grad(do_A=dA, do_S=dS, reps=20)))
        print(info["kernel"], "|", " | ".join(out), flush=True)
