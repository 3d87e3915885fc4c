#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for lib in scratch/libpmx_base.so proxmin_amd/libpmx.so; do
PMX_LIB=$PWD/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
for (M, N) in ((8192, 16384), (65536, 16384)):
    K = 128
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    dev = DeviceNMF(M, N, K, mode="f16x2")
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, "adaprox", False)
    run(10); dev.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(30); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, n = dev.get_timing()
    A, S = dev.get_factors()
    h = hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12]
    print("%s M %d: adaprox iteration %.4f ms, K1 %.4f ms, rest %.4f ms, factors sha1 %s" % (os.environ["PMX_LIB"].split("/")[-1], M, dt / 30 * 1e3, ms / max(n, 1), dt / 30 * 1e3 - ms / max(n, 1), h), flush=True)
    dev.close(); del Y; torch.cuda.empty_cache()
PY
done; done
