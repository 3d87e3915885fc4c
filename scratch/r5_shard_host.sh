#!/bin/bash
# is the sharded loop host-bound?  (one process playing rank 0 of 8, cfg4's share)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_f; mkdir -p $O
run() { echo "== $*"; env "$@" PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | host %.4f | k1 %.4f | phases %s' % (d['value'], d['ms_per_step'], d['host_enqueue_done_ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if isinstance(v,float)}))"; }
run A=1
run PMX_BENCH_CHUNK=64
run PMX_BENCH_NO_PHASES=1
run PMX_BENCH_CHUNK=64 PMX_BENCH_NO_PHASES=1
run PMX_S_SPLIT=0
run PMX_COMM=native
