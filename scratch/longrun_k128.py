# scratch: 60 iterations of pgm and adaprox on cfg4's 8192-row share (8192 x 16384 x 128): k_grad_f16_k128 (twice: bit-identical) vs exact fp32
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = 8192, 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
for backend in ("pgm", "adaprox"):
    out = {}
    for tag, mode in (("h1", "f16x2"), ("h2", "f16x2"), ("f", "f32")):
        dev = DeviceNMF(M, N, K, device=0, mode=mode)
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        run = bench.begin_solver(dev, backend, False)
        t0 = time.time(); res = run(60); dt = time.time() - t0
        A, S = dev.get_factors()
        out[tag] = (A.copy(), S.copy(), dev.loglike())
        print(backend, tag, mode, dev.k1_info()["kernel"], "60 iterations in %.2f s, loss %.6e" % (dt, out[tag][2]), flush=True)
        dev.close()
    print(backend, "f16x2 twice bit-identical:", np.array_equal(out["h1"][0], out["h2"][0]) and np.array_equal(out["h1"][1], out["h2"][1]))
    for nm, i in (("A", 0), ("S", 1)):
        a, b = out["f"][i].astype(np.float64), out["h1"][i].astype(np.float64)
        r = np.abs(a - b) / (2e-5 + 2e-4 * np.abs(a))
        print(backend, nm, "f16x2 vs f32 after 60 its: within bound %.5f, max ratio %.1f, rel Frobenius %.2e" % ((r <= 1).mean(), r.max(), np.linalg.norm(a - b) / np.linalg.norm(a)))
    print(backend, "loss rel diff %.2e" % (abs(out["f"][2] - out["h1"][2]) / out["f"][2]))
