#!/bin/bash
# [r6] the same prefetch for the ROW-SPLIT chained consumers of k_grad_f16_v8 (cfg5's A pass; mode f16x2): scratch/libpmx_pfrN.so = -DPMX_CHAIN_PFR=N builds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s; mkdir -p $O; cd $R
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3; do
  for v in 0 1 2; do
    echo -n "rep $rep pfr$v cfg5 f16x2r      : "; PMX_LIB=$R/scratch/libpmx_pfr$v.so python bench.py --config cfg5 --no-cpu --steps 60 --warmup 20 2>/dev/null | line
    echo -n "rep $rep pfr$v cfg3 mode f16x2 : "; PMX_LIB=$R/scratch/libpmx_pfr$v.so python bench.py --mode f16x2 --no-cpu --steps 100 --warmup 20 2>/dev/null | line
  done
done | tee $O/ab.txt
