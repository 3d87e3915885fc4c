"""[r6] timing-only ablations of the chain hand-off inside k_grad_f16_v8<HH, RS, CHAIN> (PMX_LIB = a -DPMX_CHAIN_ABL=n build of a scratch copy of the sources:
bit 0: no fetch / add of the previous sum, bit 1: no arrival look / wait, bit 2: the arrival published without waiting for the stores): K1 back to back on fixed
factors (pmx_time_grad), cfg3's shape.  The gradients of an ablated build are WRONG; only the time is read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = (16384, 16384, 64) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1].split("x"))
Y, A0, S0 = bench.make_problem_device(M, N, K, K == 64, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, device=0, mode="f16x2r")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
dev.time_grad(True, True, 30)
t = [dev.time_grad(True, True, 100) for _ in range(3)]
info = dev.k1_info()
print("%-28s chain %2d | K1 back to back %s ms" % (os.path.basename(os.environ.get("PMX_LIB", "default")) + (" PMX_K1_CHAIN=0" if os.environ.get("PMX_K1_CHAIN") == "0" else ""), info["chain"], " ".join("%.4f" % x for x in t)))
