# [r6] K1's launch time (HIP events on its stream, every launch) and the step time along ONE run of cfg3, in windows of 20 iterations:
# is the 20-step line of bench.py (iterations ~20..40) slower than the 100-step line (25..125) because of WHERE in the run it sits?
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, _ = bench.CONFIGS["cfg3"]
dev_t = torch.device("cuda", 0)
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, dev_t)
for rep in range(2):
    dev = DeviceNMF(M, N, K)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, backend, unity)
    seen = [0, 0]
    print("rep %d: window | it/s | K1 ms | step - K1 ms | prox passes per iteration (A, S) | fraction of zeros in A, S" % rep)
    for w in range(12):
        dev.set_timing(True, every=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run(20)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = dev.get_timing()
        dev.set_timing(False)
        sub = [(int(r.sub_iterations[j]) - seen[j]) / 20.0 for j in range(2)]
        seen = [int(r.sub_iterations[0]), int(r.sub_iterations[1])]
        zA = zS = float("nan")
        if rep == 1:          # (the download idles the GPU for a moment: rep 0 runs without it)
            A, S = dev.get_factors()
            zA, zS = float((A == 0).mean()), float((S == 0).mean())
        print("  %3d..%3d | %7.1f | %.4f | %.4f | %.2f %.2f | %.4f %.4f" % (20 * w, 20 * w + 20, 20 / dt, ms / max(n, 1), 1e3 * dt / 20 - ms / max(n, 1), sub[0], sub[1], zA, zS), flush=True)
    dev.close()
