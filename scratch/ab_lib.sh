#!/bin/bash
# same-box A/B of two builds of the library (PMX_LIB): default bench, alternating
for rep in 1 2 3; do
for lib in scratch/libpmx_base.so proxmin_amd/libpmx.so; do
  PMX_LIB=$PWD/$lib python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib it/s=%.1f ms/step=%.4f k1_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
done
