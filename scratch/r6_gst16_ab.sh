#!/bin/bash
# [r6] gSt waves of k_grad_f16_v8<HH, RS> on v_mfma_f32_16x16x32_f16 (scratch/libpmx_g16_1.so) against 32x32x16 (libpmx_g16_0.so): correctness first, then alternating bench runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_v; mkdir -p $O; cd $R
LIBS=${LIBS:-"g16_0 g16_1"}
for v in $CHECKLIBS; do echo "== $v"; PMX_LIB=$R/scratch/libpmx_$v.so python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; PMX_LIB=$R/scratch/libpmx_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3; done | tee $O/correctness.txt
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3; do
  for v in $LIBS; do
    echo -n "rep $rep $v cfg3 100/20: "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | line
    echo -n "rep $rep $v cfg3 20/5  : "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line
  done
done | tee $O/ab.txt
