#!/bin/bash
# [r6] gA waves of k_grad_f16_v8<HH, RS>: the last slot of a panel tile by tile with the flush between the MFMAs (libpmx_flt1.so) against the flush behind the slot (libpmx_flt0.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ae; mkdir -p $O; cd $R
LIBS=${LIBS:-"flt0 flt1"}
for v in $LIBS; do PMX_LIB=$R/scratch/libpmx_$v.so python scratch/r6_chain_pf_check.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -1; done | tee $O/bit_identity.txt
for v in $LIBS; do echo "== $v"; PMX_LIB=$R/scratch/libpmx_$v.so python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; PMX_LIB=$R/scratch/libpmx_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -1; done | tee $O/correctness.txt
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3 4; do
  for v in $LIBS; do
    echo -n "rep $rep $v cfg3 100/20: "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | line
    echo -n "rep $rep $v cfg3 20/5  : "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line
  done
done | tee $O/ab.txt
for v in $LIBS; do PMX_LIB=$R/scratch/libpmx_$v.so python scratch/r6_chain_ablation.py 2>&1 | grep "K1 back"; done | tee $O/b2b.txt
