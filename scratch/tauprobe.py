import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, _ = bench.CONFIGS["cfg3"]
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
with DeviceNMF(M, N, K, mode="bf16x3") as dev:
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, backend, unity)
    prev = (0, 0)
    for k in range(12):
        r = run(10)
        cur = (int(r.sub_iterations[0]), int(r.sub_iterations[1]))
        print("iterations %3d-%3d: passes per iteration A %.1f  S %.1f" % (10 * k, 10 * k + 9, (cur[0] - prev[0]) / 10, (cur[1] - prev[1]) / 10))
        prev = cur
