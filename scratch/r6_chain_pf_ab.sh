#!/bin/bash
# [r6] chain prefetch depth A/B: scratch/libpmx_pfXY.so = -DPMX_CHAIN_PF=X -DPMX_CHAIN_PF128=Y builds of the same sources (built in the container, see profiles/r06_q_*)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_q; mkdir -p $O; cd $R
LIBS=${LIBS:-"00 11 22 42"}
for v in $LIBS; do PMX_LIB=$R/scratch/libpmx_pf$v.so python scratch/r6_chain_pf_check.py 2>&1 | grep -v Warning; done | tee $O/bit_identity.txt
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3; do
  for v in $LIBS; do
    echo -n "rep $rep pf$v cfg3 20/5  : "; PMX_LIB=$R/scratch/libpmx_pf$v.so python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line
    echo -n "rep $rep pf$v cfg3 100/20: "; PMX_LIB=$R/scratch/libpmx_pf$v.so python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | line
    echo -n "rep $rep pf$v cfg4 share : "; PMX_LIB=$R/scratch/libpmx_pf$v.so python bench.py --config cfg4 --rows 8192 --no-cpu --steps 40 --warmup 20 2>/dev/null | line
    echo -n "rep $rep pf$v cfg5       : "; PMX_LIB=$R/scratch/libpmx_pf$v.so python bench.py --config cfg5 --no-cpu --steps 40 --warmup 10 2>/dev/null | line
  done
done | tee $O/ab.txt
