#!/bin/bash
# default bench: the working tree's library against the previous commit's; same box, alternating; then the tail's phase stamps
cd $GRAFT_REPO_ROOT
{
for rep in 1 2 3 4; do
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  PMX_LIB=$PWD/$lib python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done
done
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  echo $lib; PMX_LIB=$PWD/$lib PMX_TAIL_PROF=1 python bench.py --steps 40 --warmup 10 --no-cpu 2>&1 | grep tailprof | tail -1
done
} | tee gpurun_out/r4_tail_ab.txt
