#!/bin/bash
# round 3: k_grad_f16_k128<.., FOLD> -- tests, then same-box A/B of cfg4's 8192-row share and of cfg4 on one GPU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_fold
O=gpurun_out/r3_fold
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "k128" > $O/k_tests.txt 2>&1; echo "kernels rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/k_tests.txt | head -20
timeout 600 python - <<'PY' 2>&1 | tee $O/ab.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
for (M, N) in ((8192, 16384), (65536, 16384)):
    K = 128
    Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    for fold in ("1", "0", "1", "0"):
        os.environ["PMX_K1_K128_FOLD"] = fold
        dev = DeviceNMF(M, N, K, mode="f16x2")
        dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
        dev.set_factors(A0, S0)
        info = dev.k1_info()
        t11 = dev.time_grad(1, 1, 30)
        run = bench.begin_solver(dev, "adaprox", False)
        run(10); dev.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(30); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, n = dev.get_timing()
        print("M %d fold %s (info fold=%s slabs_A=%d): K1 back-to-back %.4f ms | adaprox iteration %.4f ms, K1 %.4f ms, faults %s" % (
            M, fold, info["fold"], info["slabs_A"], t11, dt / 30 * 1e3, ms / max(n, 1), dev.k1_info()["chain_faults"]), flush=True)
        dev.close()
    del Y
    torch.cuda.empty_cache()
PY
