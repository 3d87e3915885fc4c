#!/bin/bash
# L2 <-> fabric traffic of the K = 128 K1, slabs against chains: which requests make up WRITE_SIZE
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_pmc_k128; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
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
for var in chain slab; do
  [ $var = slab ] && export PMX_K1_CHAIN=0 || unset PMX_K1_CHAIN
  for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_INV_EVICT_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum"; do
    rm -rf $O/p
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p -o p -- python /tmp/k128_one.py > /dev/null 2> $O/err.txt
    python - "$var" "$C" $O/p <<'PY'
import csv, glob, sys, collections
var, C, d = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(var, C, "no output"); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_grad_f16_k128" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-6s %-34s mean per launch %.4g (n=%d)" % (var, k, sum(v) / len(v), len(v)))
PY
  done
done 2>&1 | tee $O/counters.txt
