# scratch: what does ONE pmx_adaprox_run(n) call cost beyond n iterations?  cfg3, mode f16x2r: wall time of run(n) for n = 1 .. 100 (intercept of the line) and the K1 launch times
# of the first iterations after an idle stream (HIP events around every K1).
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, _ = bench.CONFIGS["cfg3"]
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, device=0, mode="f16x2r")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
run = bench.begin_solver(dev, backend, unity)
run(60)
for rep in range(2):
    for n in (1, 2, 4, 8, 20, 40, 100):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("run(%3d): %.3f ms total, %.4f ms per iteration" % (n, dt * 1e3, dt * 1e3 / n), flush=True)
for n in (1, 2, 4, 8, 20):
    dev.set_timing(True, every=1)
    torch.cuda.synchronize(); run(n); torch.cuda.synchronize()
    ms, k = dev.get_timing(); dev.set_timing(False)
    print("K1 averaged over the %d launches of run(%d): %.4f ms" % (k, n, ms / max(k, 1)), flush=True)
for idle in (0.0, 0.001, 0.01, 0.1):
    dev.set_timing(True, every=1)
    torch.cuda.synchronize(); time.sleep(idle); run(2); torch.cuda.synchronize()
    ms, k = dev.get_timing(); dev.set_timing(False)
    print("after %.0f ms of idle: K1 over run(2) %.4f ms" % (idle * 1e3, ms / max(k, 1)), flush=True)
dev.close()
