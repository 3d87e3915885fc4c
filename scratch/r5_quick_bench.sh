#!/bin/bash
# quick lines: cfg2 (f16x2r, f32), cfg3 (f16x2r), cfg5, each --no-cpu
cd $GRAFT_REPO_ROOT
for spec in "cfg2 f16x2r 300 40" "cfg2 f16x2 300 40" "cfg2 f32 300 40" "cfg3 f16x2r 100 20" "cfg5 f16x2r 40 10"; do
  set -- $spec
  python bench.py --config $1 --mode $2 --steps $3 --warmup $4 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 it/s %.1f ms %.4f k1 %.4f tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done
