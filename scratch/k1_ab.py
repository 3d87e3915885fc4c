# scratch: K1 variants A/B on bench data (cfg3), back to back and inside the iteration chain
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
device = torch.device("cuda", 0)
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, device)
for rep in range(2):
    for variant in sys.argv[1:]:
        os.environ["PMX_K1_VARIANT"] = variant
        dev = DeviceNMF(M, N, K, device=0, mode=os.environ.get("PMX_AB_MODE", "bf16x3"))
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, backend, unity)
        run(20)
        dev.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(200)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, n = dev.get_timing(); dev.set_timing(False)
        b2b = dev.time_grad(1, 1, 200)
        print("variant %s: iteration %.3f ms (%.0f it/s), K1 in chain %.3f ms, back to back %.3f ms" % (variant, dt / 200 * 1e3, 200 / dt, ms / max(n, 1), b2b), flush=True)
        dev.close()
