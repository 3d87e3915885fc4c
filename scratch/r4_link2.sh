#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_link; mkdir -p $O
for rep in 1 2 3 4; do
for lib in proxmin_amd/libpmx.so scratch/libpmx_head.so; do
  PMX_LIB=$R/$lib python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg3 it/s=%.1f ms/step=%.4f k1_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
done 2>&1 | tee $O/ab2.txt
