# scratch (round 3; [r6] the library default mode f16x2r for cfg3 / cfg5 / cfg2h): soak -- run the chained kernels past 2^21 launches (the arrival words' reset path, pmx_api.hip) and check
# that nothing faults, nothing goes non-finite, and the result equals a second run's bit for bit
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
which = sys.argv[1]
total = int(sys.argv[2])
cfg = {"cfg2": (4096, 4096, 32, "pgm", "f32", False), "cfg3": (16384, 16384, 64, "adaprox", "f16x2r", True),
       "cfg5": (16384, 16384, 64, "bsdmm", "f16x2r", False), "cfg2h": (4096, 4096, 32, "pgm", "f16x2r", False),      # [r4]
       "ragged": (16000, 16000, 64, "adaprox", "f16x2", True)}[which]
M, N, K, backend, mode, unity = cfg
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
finals = []
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    dev = DeviceNMF(M, N, K, mode=mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, backend, unity)
    done, t0 = 0, time.perf_counter()
    step = 100000
    while done < total:
        n = min(step, total - done)
        r = run(n)
        if r.iterations < n:             # pgm at e_rel = 1e-12 does converge (fp32: the iterate stops moving): start over, same context
            print("   converged after %d of %d: restart from the initial factors" % (r.iterations, n), flush=True)
            dev.set_factors(A0, S0)
            run = bench.begin_solver(dev, backend, unity)
        done += max(int(r.iterations), 1)
        info = dev.k1_info()
        loss = dev.loglike()
        print("%s rep %d: %8d iterations, %.0f s, loss %.6e, chain %d, faults %d / %d, fused %s" % (
            which, rep, done, time.perf_counter() - t0, loss, info["chain"], info["chain_faults"], info["tail_faults"], info["tail_fused"]), flush=True)
        assert np.isfinite(loss) and info["chain_faults"] == 0 and info["tail_faults"] == 0
    A, S = dev.get_factors()
    assert np.isfinite(A).all() and np.isfinite(S).all()
    finals.append((A, S))
    dev.close()
if len(finals) == 2:
    print("bitwise equal runs:", np.array_equal(finals[0][0], finals[1][0]) and np.array_equal(finals[0][1], finals[1][1]))
print("soak ok")
