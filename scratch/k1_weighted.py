# scratch: K1 with an M x N weight array, split-bf16 (v7 weighted instance) vs exact fp32
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
W = torch.rand((M, N), device="cuda")
for mode in ("bf16x3", "f32"):
    dev = DeviceNMF(M, N, K, device=0, mode=mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    t0 = dev.time_grad(1, 1, 50)
    dev.set_W_device(W.data_ptr(), ld=N, copy=False, keepalive=W)
    t1 = dev.time_grad(1, 1, 50)
    print("%s: K1 unweighted %.3f ms, weighted %.3f ms (back to back)" % (mode, t0, t1), flush=True)
    dev.close()
