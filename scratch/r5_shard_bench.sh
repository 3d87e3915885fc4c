#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_e; mkdir -p $O
python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg4_1gpu.json 2>/dev/null
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg4_shard8192_rank0of8.json 2>/dev/null
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --rows 2048 --steps 100 --warmup 20 --no-cpu > $O/bench_shard2048_rank0of8.json 2>/dev/null
for f in $O/bench_*.json; do echo "== $f"; grep '^{' $f | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | phases %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('phases_ms')))"; done
