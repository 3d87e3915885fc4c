#!/bin/bash
# round 4: the whole GPU suite + the default bench line + one rank-0-of-8 sharded line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_full; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -15 $O/tests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 3000 $O/bench_default.json
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 timeout 300 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg4_shard8192_rank0of8.json 2> $O/bench_shard.err; tail -c 1500 $O/bench_cfg4_shard8192_rank0of8.json
PMX_FORCE_SHARDED=1 timeout 300 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg4_shard8192_world1_native.json 2> $O/bench_shard_w1.err; tail -c 800 $O/bench_cfg4_shard8192_world1_native.json; tail -3 $O/bench_shard_w1.err
cp gpurun_out/parity_fractions.json $O/ 2>/dev/null
