#!/bin/bash
# [r6] the correction's three launches: DevStatus::halt requested at the top and consulted BEHIND the first loads (libpmx_halt1.so) against the check in front of everything (libpmx_halt0.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_x; mkdir -p $O; cd $R
LIBS=${LIBS:-"halt0 halt1"}
for v in $LIBS; do PMX_LIB=$R/scratch/libpmx_$v.so python scratch/r6_chain_pf_check.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/bit_identity.txt
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3 4; do
  for v in $LIBS; do
    echo -n "rep $rep $v cfg3 100/20: "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | line
    echo -n "rep $rep $v cfg4 share : "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --config cfg4 --rows 8192 --no-cpu --steps 40 --warmup 20 2>/dev/null | line
    echo -n "rep $rep $v cfg5       : "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --config cfg5 --no-cpu --steps 60 --warmup 20 2>/dev/null | line
  done
done | tee $O/ab.txt
for v in $LIBS; do echo "== $v"; PMX_LIB=$R/scratch/libpmx_$v.so PMX_GFIX_PROF=1 python scratch/r6_gfix_prof.py 2>&1 | grep gfixprof; done | tee $O/gfix_stamps.txt
