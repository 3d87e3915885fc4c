#!/bin/bash
# round 4: chained K = 128 K1, second pass -- flush-before-gSt A/B, fault tests, timeline + PMC traffic of the share
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=$R/gpurun_out/r4_k128b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nmf.py -q -x -k "k128 or (chained and 128)" > $O/k_tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/k_tests.txt | head -20
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
  TAG=new-c16s1 python /tmp/k128_ab.py 2>&1 | tail -1
  TAG=new-c32s1 PMX_K1_CHAIN=32 python /tmp/k128_ab.py 2>&1 | tail -1
done 2>&1 | tee $O/ab.txt
ROWS=65536 TAG=full-new python /tmp/k128_ab.py 2>&1 | tail -1 | tee -a $O/ab.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 timeout 300 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg4_shard8192_rank0of8.json 2> $O/bench_shard.err
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/trace.json 2> $O/trace.err
python scratch/trace_gaps.py $(ls $O/trace/*kernel_trace.csv) 20000 > $O/timeline_cfg4_shard8192_ssplit.txt
cp $(ls $O/trace/*kernel_stats.csv) $O/kernel_stats_cfg4_shard8192.csv
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python /tmp/k128_ab.py > $O/trace1.txt 2> $O/trace1.err
python scratch/trace_gaps.py $(ls $O/trace/*kernel_trace.csv) 20000 > $O/timeline_cfg4_share_1gpu_path.txt
rm -rf $O/trace
head -40 $O/timeline_cfg4_shard8192_ssplit.txt
head -30 $O/timeline_cfg4_share_1gpu_path.txt
bash scratch/measure_traffic.sh cfg4 f16x2 8192
