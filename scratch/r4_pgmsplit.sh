#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_pgmsplit; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_distributed_world2.py tests/test_gpu_distributed.py tests/test_gpu_bench_multirank.py -q -k "pgm or fista or bench" > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests.txt | head -20
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 timeout 300 python bench.py --config cfg2 --rows 512 --steps 100 --warmup 20 --no-cpu 2> $O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['parallelism'], d['phases_ms'])"
tail -3 $O/err.txt
