import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, device=0, mode="f16x2")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
print(os.environ.get("PMX_LIB", "base")[-16:], "producers only %.3f ms | full %.3f ms" % (dev.time_grad(0, 0, 300), dev.time_grad(1, 1, 300)), flush=True)
