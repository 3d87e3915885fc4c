#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 1 2 4; do
  PMX_K128_STRIDE=$S python bench.py --config cfg4 --rows 8192 --steps 60 --warmup 20 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stride $S: it/s %.1f ms %.4f k1 %.4f tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done
