#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if isinstance(v,float)}))"; }
for i in 1 2; do
echo "== sharded path"; PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
echo "== sharded path, no k_absmax (stale maxima: experiment)"; PMX_X_STALE_ABSMAX=1 PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
done
