# scratch: race hunt -- repeated launches of K1 on the same inputs must be bit-identical (any ordering hazard between the
# LDS-DMA landings, the rings and the waves' reads shows up as run-to-run differences)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for (M, N) in ((16384, 16384), (2048, 16384), (5120, 4096), (128, 256)):
    K = 64
    Y, A0, S0 = bench.make_problem_device(M, N, K, True, 1234, torch.device("cuda", 0))
    for mode in ("f16x2", "bf16x3", "f32"):
        dev = DeviceNMF(M, N, K, device=0, mode=mode)
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        ref = None; bad = 0
        for r in range(reps):
            gA, gS = dev.grad(); loss = dev.loglike()
            cur = (gA.copy(), gS.copy(), loss)
            if ref is None: ref = cur
            elif not (np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1]) and ref[2] == cur[2]): bad += 1
        print("%dx%d %s: %d of %d repeats differ" % (M, N, mode, bad, reps - 1), flush=True)
        dev.close()
