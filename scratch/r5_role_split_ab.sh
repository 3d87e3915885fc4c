#!/bin/bash
# K1 <RS> (consumer roles split by contraction) against the row split: kernel tests, then alternating bench lines on one box
#   usage: scratch/r5_role_split_ab.sh [cfg3|cfg4]
cd $GRAFT_REPO_ROOT
CFG=${1:-cfg3}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gfix.py -m gpu -x -q 2>&1 | tail -3
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms') or 0))"; }
if [ $CFG = cfg3 ]; then ARGS="--steps 100 --warmup 20"; else ARGS="--config cfg4 --rows 8192 --steps 60 --warmup 20"; fi
for i in 1 2 3; do
echo -n "RS=1 "; python bench.py $ARGS --no-cpu 2>/dev/null | line
echo -n "RS=0 "; PMX_K1_ROLE_SPLIT=0 python bench.py $ARGS --no-cpu 2>/dev/null | line
done
