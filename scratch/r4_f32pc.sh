#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_f32pc; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_strict.py tests/test_gpu_parity_long.py -q -k "f32 or producer_consumer or chained or cfg2 or fused_gradient" > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests.txt | head
for rep in 1 2 3; do
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  PMX_LIB=$R/$lib python bench.py --config cfg2 --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg2 f32 it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done
done 2>&1 | tee $O/ab.txt
for lib in scratch/libpmx_head.so proxmin_amd/libpmx.so; do
  PMX_LIB=$R/$lib python bench.py --no-cpu --mode f32 --steps 40 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg3 f32 it/s=%.1f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done 2>&1 | tee -a $O/ab.txt
