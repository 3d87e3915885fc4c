#!/bin/bash
# round 4: chained K = 128 K1 -- parity tests, then same-box A/B (r3 library / new slabs / new chain)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_k128; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "k128 or chained or operand_scaling" > $O/k_tests.txt 2>&1; echo "kernels rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/k_tests.txt | head -20
cat > /tmp/k128_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = int(os.environ.get("ROWS", 8192)), 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, mode="f16x2")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
info = dev.k1_info()
t_all, t_res, t_a, t_s = dev.time_grad(1, 1, 100), dev.time_grad(0, 0, 100), dev.time_grad(1, 0, 100), dev.time_grad(0, 1, 100)
run = bench.begin_solver(dev, "adaprox", False)
run(20); dev.set_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
ms, n = dev.get_timing()
print("%s chain=%s slabsA=%d slabsS=%d | K1 b2b %.4f (res %.4f, +gA %.4f, +gS %.4f) | adaprox it %.4f ms K1 %.4f ms faults %s" % (
    os.environ.get("TAG"), info["chain"], info["slabs_A"], info["slabs_S"], t_all, t_res, t_a, t_s, dt / 60 * 1e3, ms / max(n, 1), dev.k1_info()["chain_faults"]))
dev.close()
PY
for rep in 1 2; do
  TAG=r3      PMX_LIB=$PWD/scratch/libpmx_r3.so python /tmp/k128_ab.py 2>&1 | tail -1
  TAG=new-slab PMX_K1_CHAIN=0 python /tmp/k128_ab.py 2>&1 | tail -1
  TAG=new-c16s2 python /tmp/k128_ab.py 2>&1 | tail -1
  TAG=new-c16s1 PMX_K128_STRIDE=1 python /tmp/k128_ab.py 2>&1 | tail -1
  TAG=new-c32 PMX_K1_CHAIN=32 python /tmp/k128_ab.py 2>&1 | tail -1
  TAG=new-c8 PMX_K1_CHAIN=8 python /tmp/k128_ab.py 2>&1 | tail -1
done 2>&1 | tee $O/ab.txt
ROWS=65536 TAG=full-r3 PMX_LIB=$PWD/scratch/libpmx_r3.so python /tmp/k128_ab.py 2>&1 | tail -1 | tee -a $O/ab.txt
ROWS=65536 TAG=full-new python /tmp/k128_ab.py 2>&1 | tail -1 | tee -a $O/ab.txt
