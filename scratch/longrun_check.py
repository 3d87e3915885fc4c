# scratch: 300 adaprox iterations at cfg3, twice in split-bf16 mode (must be bit-identical) and once in exact fp32
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
out = {}
for tag, mode in (("b1", sys.argv[1] if len(sys.argv) > 1 else "bf16x3"),) * 1 + (("b2", sys.argv[1] if len(sys.argv) > 1 else "bf16x3"), ("f", "f32")):
    dev = DeviceNMF(M, N, K, device=0, mode=mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, backend, unity)
    t0 = time.time(); res = run(300); dt = time.time() - t0
    A, S = dev.get_factors()
    out[tag] = (A.copy(), S.copy(), dev.loglike())
    print(tag, mode, "300 iterations in %.2f s, loss %.6e, sub-iterations %s" % (dt, out[tag][2], res.sub_iterations), flush=True)
    dev.close()
print("bf16x3 twice bit-identical:", np.array_equal(out["b1"][0], out["b2"][0]) and np.array_equal(out["b1"][1], out["b2"][1]))
for nm, i in (("A", 0), ("S", 1)):
    a, b = out["f"][i].astype(np.float64), out["b1"][i].astype(np.float64)
    r = np.abs(a - b) / (2e-5 + 2e-4 * np.abs(a))
    print(nm, "bf16x3 vs f32 after 300 its: within bound %.5f, >25x %.2e, max ratio %.1f, rel Frobenius %.2e" % ((r <= 1).mean(), (r > 25).mean(), r.max(), np.linalg.norm(a - b) / np.linalg.norm(a)))
print("loss rel diff %.2e" % (abs(out["f"][2] - out["b1"][2]) / out["f"][2]))
