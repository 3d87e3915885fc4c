#!/bin/bash
# same-box A/B of library builds: default bench (no CPU leg), alternating; LIBS = space-separated .so paths
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_ab
LIBS=${LIBS:-"scratch/libpmx_r2.so scratch/libpmx_s1.so proxmin_amd/libpmx.so"}
for rep in 1 2 3; do
for lib in $LIBS; do
  PMX_LIB=$PWD/$lib python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib it/s=%.1f ms/step=%.4f k1_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
done | tee gpurun_out/r3_ab/ab.txt
