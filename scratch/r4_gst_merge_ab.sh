#!/bin/bash
# k_grad_f16_v8 / k_grad_bf16_v7: gSt row halves merged inside the launch (one slab per row region), against the previous commit; same box, alternating
cd $GRAFT_REPO_ROOT
one() { # lib config mode steps warmup
  PMX_LIB=$PWD/$1 python bench.py --config $2 ${3:+--mode $3} --steps $4 --warmup $5 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 $3 it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms', float('nan'))))"
}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nmf.py -q -x 2>&1 | tail -3
{
for rep in 1 2 3; do
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  one $lib cfg3 "" 100 20
  one $lib cfg3 bf16x3 60 15
  one $lib cfg5 "" 60 15
done
done
} | tee gpurun_out/r4_gst_merge_ab.txt
