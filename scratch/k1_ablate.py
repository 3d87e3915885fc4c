import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from proxmin_amd.engine import DeviceNMF
M = N = int(os.environ.get("SZ", 16384)); K = int(os.environ.get("K", 64))
Y = torch.rand((M, N), device="cuda")
rng = np.random.default_rng(0)
A0 = rng.random((M, K), dtype=np.float32); S0 = rng.random((K, N), dtype=np.float32)
for mode in sys.argv[1:] or ["f32", "bf16x3"]:
    dev = DeviceNMF(M, N, K, mode=mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    for doA, doS in ((1, 1), (0, 0), (1, 0), (0, 1), (2, 0), (3, 1)):
        ms = dev.time_grad(doA, doS, 10)
        print("%s doA=%d doS=%d: %.3f ms  (%.0f GB/s Y, %.1f TFLOP/s alg)" % (mode, doA, doS, ms, M * N * 4 / ms / 1e6, (2 + 2 * doA + 2 * doS) * M * N * K / ms / 1e9))
    dev.close()
