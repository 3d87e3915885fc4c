#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2 3; do
echo -n "fresh accumulator "; python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | line
echo -n "launch-long only  "; PMX_LIB=$PWD/scratch/libpmx_base.so python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | line
done
