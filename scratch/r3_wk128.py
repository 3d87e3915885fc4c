# scratch (round 3): weighted K = 128 K1: k_grad_f16_k128<HASW> against the exact-fp32 kernel (the former fall-back), timing
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = 8192, 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
W = 0.1 + 2.0 * torch.rand((M, N), device="cuda")
for mode in ("f16x2", "f32"):
    dev = DeviceNMF(M, N, K, mode=mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    t0 = dev.time_grad(1, 1, 50)
    dev.set_W_device(W.data_ptr(), ld=N, copy=False, keepalive=W)
    print(mode, dev.k1_info()["kernel"], "unweighted %.4f ms, weighted %.4f ms" % (t0, dev.time_grad(1, 1, 50)), flush=True)
    dev.close()
