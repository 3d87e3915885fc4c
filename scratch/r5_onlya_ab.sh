#!/bin/bash
# experiment: the gA-only pass (bsdmm's A step) on the <RS> kernel (two waves contract gA, the gSt waves idle) against the row split
cd $GRAFT_REPO_ROOT
PMX_X_RS_ONLYA=1 timeout 600 python -m pytest tests/test_gpu_gfix.py -m gpu -x -q 2>&1 | tail -2
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2 3; do
echo -n "A pass on <RS>   "; PMX_X_RS_ONLYA=1 python bench.py --config cfg5 --steps 60 --warmup 10 --no-cpu 2>/dev/null | line
echo -n "A pass row split "; python bench.py --config cfg5 --steps 60 --warmup 10 --no-cpu 2>/dev/null | line
done
