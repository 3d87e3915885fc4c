#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_k128
O=gpurun_out/r3_k128
true || timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "k128" > $O/k_tests.txt 2>&1; echo "kernels rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/k_tests.txt | head -10
true || timeout 900 python -m pytest tests/test_gpu_parity_long.py -q -k "k128 or cfg4" > $O/long_tests.txt 2>&1; echo "long rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/long_tests.txt | head -10
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g; g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = 8192, 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
for chain in ("32", "0"):
    os.environ["PMX_K1_CHAIN"] = chain
    dev = DeviceNMF(M, N, K, mode="f16x2")
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    print("chain", chain, dev.k1_info())
    print("   K1 back-to-back: %.4f ms (gA+gS), %.4f (residual only)" % (dev.time_grad(1, 1, 100), dev.time_grad(0, 0, 100)))
    if chain != "0":
        print("   ABLATION no fold at all: %.4f ms; fold without previous sums / waits: %.4f ms; doA only %.4f" % (dev.time_grad(1 | 256, 1, 100), dev.time_grad(1 | 512, 1, 100), dev.time_grad(1, 0, 100)))
    run = bench.begin_solver(dev, "adaprox", False)
    run(20); dev.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, n = dev.get_timing()
    print("   adaprox iteration %.4f ms, K1 %.4f ms, faults %s" % (dt / 40 * 1e3, ms / max(n, 1), dev.k1_info()["chain_faults"]))
    dev.close()
PY
