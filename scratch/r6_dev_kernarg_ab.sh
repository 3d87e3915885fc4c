#!/bin/bash
# [r6] HIP_FORCE_DEV_KERNARG: kernel arguments in device memory instead of host memory (every kernel's first instructions read them)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_aa; mkdir -p $O; cd $R
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "rep $rep HIP_FORCE_DEV_KERNARG=$v cfg3 100/20: "; HIP_FORCE_DEV_KERNARG=$v python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | line
    echo -n "rep $rep HIP_FORCE_DEV_KERNARG=$v cfg2 f32    : "; HIP_FORCE_DEV_KERNARG=$v python bench.py --config cfg2 --no-cpu --steps 400 --warmup 50 2>/dev/null | line
    echo -n "rep $rep HIP_FORCE_DEV_KERNARG=$v cfg2 f16x2r : "; HIP_FORCE_DEV_KERNARG=$v python bench.py --config cfg2 --mode f16x2r --no-cpu --steps 400 --warmup 50 2>/dev/null | line
    echo -n "rep $rep HIP_FORCE_DEV_KERNARG=$v cfg5        : "; HIP_FORCE_DEV_KERNARG=$v python bench.py --config cfg5 --no-cpu --steps 60 --warmup 20 2>/dev/null | line
  done
done | tee $O/ab.txt
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v PMX_GFIX_PROF=1 python scratch/r6_gfix_prof.py 2>&1 | grep gfixprof | head -3; done | tee $O/gfix_stamps.txt
