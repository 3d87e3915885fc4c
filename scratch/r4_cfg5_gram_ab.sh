#!/bin/bash
# cfg5 (bsdmm) and cfg2 (pgm): partial Gram matrices out of k_bsdmm_update + 256 Gram shares, against the previous commit's library; same box, alternating
cd $GRAFT_REPO_ROOT
one() { # lib config mode steps warmup
  PMX_LIB=$PWD/$1 python bench.py --config $2 ${3:+--mode $3} --steps $4 --warmup $5 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 $3 it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms', float('nan'))))"
}
{
for rep in 1 2 3; do
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  one $lib cfg5 "" 60 15
  one $lib cfg2 f32 300 40
  one $lib cfg2 f16x2 300 40
done
done
PMX_GRAM_IN_UPDATE=0 python bench.py --config cfg5 --steps 60 --warmup 15 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new library, PMX_GRAM_IN_UPDATE=0: cfg5 it/s=%.1f' % d['value'])"
} | tee gpurun_out/r4_cfg5_gram_ab.txt
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_callbacks.py tests/test_gpu_parity_strict.py -q -x 2>&1 | tail -5
