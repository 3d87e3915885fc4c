#!/bin/bash
# why is K1 8 % slower (HIP events) in the sharded code path than in the single-GPU one at the same shape (cfg4's share), when rocprofv3 sees it 3 % FASTER there?
cd $GRAFT_REPO_ROOT
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v,4) for k,v in d.get('phases_ms',{}).items() if isinstance(v,float)}))"; }
for i in 1 2; do
echo "== single-GPU path"; python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
echo "== sharded path (torch stream)"; PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
echo "== sharded path, the library's own stream"; PMX_BENCH_OWN_STREAM=1 PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
echo "== sharded path, no phase events"; PMX_BENCH_NO_PHASES=1 PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
echo "== sharded path, S replicated"; PMX_S_SPLIT=0 PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 64 --warmup 10 --no-cpu 2>/dev/null | line
done
