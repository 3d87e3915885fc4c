#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_fold
for lib in proxmin_amd/libpmx.so scratch/libpmx_abl1.so scratch/libpmx_abl2.so; do
PMX_LIB=$PWD/$lib timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = 8192, 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
for fold in ("1", "0"):
    os.environ["PMX_K1_K128_FOLD"] = fold
    dev = DeviceNMF(M, N, K, mode="f16x2")
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    t11 = dev.time_grad(1, 1, 50)
    t10 = dev.time_grad(1, 0, 50)
    print("%s fold %s: K1 back-to-back %.4f ms (gA only %.4f)" % (os.environ["PMX_LIB"].split("/")[-1], fold, t11, t10), flush=True)
    dev.close()
PY
done | tee gpurun_out/r3_fold/abl.txt
