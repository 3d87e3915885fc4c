# scratch (round 3): cfg2's K1 (k_grad_f32_pc<32>, 4096^2): fixed cost against per-block cost -- rows scaled at N = 4096
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
N, K = 4096, 32
for mode in ("f32", "f16x2"):
    for M in (4096, 8192, 16384, 32768):
        Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
        dev = DeviceNMF(M, N, K, mode=mode)
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        info = dev.k1_info()
        t = [dev.time_grad(a, s, 200) * 1e3 for a, s in ((1, 1), (0, 0), (1, 0), (0, 1))]
        print("%s M %5d %s chain %d RP %d grid %dx%d: all %.1f us | residual only %.1f | +gA %.1f | +gSt %.1f" % (
            mode, M, info["kernel"], info["chain"], info["panels_per_region"], info["row_regions"], info["col_regions"], *t), flush=True)
        dev.close(); del Y
