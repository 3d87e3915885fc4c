# would zero-padding K = 32 to 64 (two-term fp16 kernel) beat the exact-fp32 K1 at cfg2's size?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import torch, bench
from proxmin_amd import engine
M = N = 4096
for K, mode in ((32, "f32"), (32, "f16x2"), (64, "f16x2"), (64, "f32")):
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    with engine.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        print(K, mode, dev.k1_info()["kernel"], "chain", dev.k1_info()["chain"], "%.4f ms" % dev.time_grad(do_A=1, do_S=1, reps=50), flush=True)
