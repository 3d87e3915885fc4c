#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_cfg2c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests.txt | head
for rep in 1 2 3; do
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  for m in f32 f16x2; do
  PMX_LIB=$R/$lib python bench.py --config cfg2 --mode $m --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg2 $m it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
  done
done
done 2>&1 | tee $O/ab.txt
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  PMX_LIB=$R/$lib python bench.py --no-cpu --config cfg5 --steps 40 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg5 it/s=%.1f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done 2>&1 | tee -a $O/ab.txt
