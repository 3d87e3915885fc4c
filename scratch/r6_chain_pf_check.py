"""[r6] A/B helper of the chain prefetch depth (PMX_CHAIN_PF / PMX_CHAIN_PF128 builds, PMX_LIB picks the library): 12 adaprox iterations at
cfg3's and at cfg4-share's shape class with the chained K1s; prints the kernel, the chain length and a hash of the factors -- equal hashes across
the builds = the same sums in the same order, bit for bit."""
import hashlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from proxmin_amd.engine import DeviceNMF
for (M, N, K) in ((16384, 16384, 64), (8192, 16384, 128)):
    Y, A0, S0 = bench.make_problem_device(M, N, K, True, 1234, torch.device("cuda", 0))
    dev = DeviceNMF(M, N, K, device=0, mode="f16x2r")
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, "adaprox", True)
    run(12)
    A, S = dev.get_factors()
    info = dev.k1_info()
    print(os.environ.get("PMX_LIB", "default"), (M, N, K), info["kernel"], "chain", info["chain"], hashlib.sha256(np.ascontiguousarray(A).tobytes() + np.ascontiguousarray(S).tobytes()).hexdigest()[:16])
    dev.close()
    del Y
    torch.cuda.empty_cache()
