# scratch: K1 timing for an arbitrary shape / mode:  python scratch/k1_shape.py M N K mode
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from proxmin_amd.engine import DeviceNMF
M, N, K = (int(x) for x in sys.argv[1:4]); mode = sys.argv[4]
Y = torch.rand((M, N), device="cuda")
rng = np.random.default_rng(0)
A0 = rng.random((M, K), dtype=np.float32); S0 = rng.random((K, N), dtype=np.float32)
dev = DeviceNMF(M, N, K, mode=mode)
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
for _ in range(3):
    ms = dev.time_grad(1, 1, 50)
print("%dx%d K=%d %s: %.3f ms  %.1f TFLOP/s (6MNK)  %.0f GB/s Y" % (M, N, K, mode, ms, 6.0 * M * N * K / ms / 1e9, M * N * 4 / ms / 1e6))
