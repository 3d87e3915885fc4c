#!/bin/bash
# cache policy of the Y stream in the K = 128 K1 (nt / sc1 / sc0 sc1 / plain): time + L2 write traffic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_yscope; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
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
cat > /tmp/k128_one.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = 8192, 16384, 128
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, mode="f16x2")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
print(dev.k1_info(), dev.time_grad(1, 1, 12))
dev.close()
PY

for rep in 1 2; do
for v in nt y1 y2 y3; do
  [ $v = nt ] && L=$R/proxmin_amd/libpmx.so || L=$R/scratch/libpmx_$v.so
  TAG=$v PMX_LIB=$L python /tmp/k128_ab.py 2>&1 | tail -1
done
done 2>&1 | tee $O/ab.txt
for v in nt y1 y2 y3; do
  [ $v = nt ] && L=$R/proxmin_amd/libpmx.so || L=$R/scratch/libpmx_$v.so
  for C in "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf $O/p
    PMX_LIB=$L timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p -o p -- python /tmp/k128_one.py > /dev/null 2> $O/err.txt
    python - "$v" $O/p <<'PY'
import csv, glob, sys, collections
var, d = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_grad_f16_k128" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-4s %-28s mean per launch %.4g" % (var, k, sum(v) / len(v)))
PY
  done
done 2>&1 | tee $O/counters.txt
rm -rf $O/p
