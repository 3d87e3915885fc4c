#!/bin/bash
# [r6] the correction's gram kernel launched WITHOUT the barrier bit behind K1 (PMX_GFIX_ANYORDER=1: hipExtAnyOrderLaunch) against the plain launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ab; mkdir -p $O; cd $R
for v in 0 1; do PMX_GFIX_ANYORDER=$v python scratch/r6_chain_pf_check.py 2>&1 | grep -v "Warning\|amdgpu.ids" | sed "s/^/anyorder=$v /"; done | tee $O/bit_identity.txt
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "rep $rep anyorder=$v cfg3 100/20: "; PMX_GFIX_ANYORDER=$v python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | line
    echo -n "rep $rep anyorder=$v cfg3 20/5  : "; PMX_GFIX_ANYORDER=$v python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line
    echo -n "rep $rep anyorder=$v cfg4 share : "; PMX_GFIX_ANYORDER=$v python bench.py --config cfg4 --rows 8192 --no-cpu --steps 40 --warmup 20 2>/dev/null | line
    echo -n "rep $rep anyorder=$v cfg5       : "; PMX_GFIX_ANYORDER=$v python bench.py --config cfg5 --no-cpu --steps 60 --warmup 20 2>/dev/null | line
  done
done | tee $O/ab.txt
for v in 0 1; do
  rm -rf $O/kt$v
  PMX_GFIX_ANYORDER=$v rocprofv3 --kernel-trace --output-format csv -d $O/kt$v -o s -- python bench.py --steps 40 --warmup 20 --no-cpu > /dev/null 2>&1
  echo "== anyorder=$v"; python scratch/trace_gaps.py $(ls $O/kt$v/*kernel_trace.csv | head -1) 2>&1 | tail -16
  rm -rf $O/kt$v
done | tee $O/timeline.txt
