# [r6] where the three correction launches spend their time (PMX_GFIX_PROF stamps of workgroup 0, printed by pmx_time_grad)
import os, sys
os.environ["PMX_GFIX_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
for M, N, K in ((16384, 16384, 64), (8192, 16384, 128)):
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    with DeviceNMF(M, N, K) as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        for _ in range(3):
            print("%d x %d x %d: K1 + correction %.4f ms" % (M, N, K, dev.time_grad(True, True, 10)), flush=True)
    del Y
