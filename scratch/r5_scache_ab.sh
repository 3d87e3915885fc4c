#!/bin/bash
# <RS>: the gA waves keep the S fragments of the region's first n blocks in registers (PMX_RS_SCACHE=n builds: scratch/libpmx_sc<n>.so) against n = 0
cd $GRAFT_REPO_ROOT
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
PMX_LIB=$PWD/scratch/libpmx_sc3.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "f16x2r" 2>&1 | tail -2
for i in 1 2 3; do
for n in 0 2 3; do
  if [ $n = 0 ]; then L=$PWD/proxmin_amd/libpmx.so; else L=$PWD/scratch/libpmx_sc$n.so; fi
  echo -n "NSC=$n "; PMX_LIB=$L python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | line
done
done
