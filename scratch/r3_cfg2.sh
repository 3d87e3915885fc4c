#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_cfg2
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nmf.py -q -x -k "step or pgm or fista or eig or lambda or bsdmm or backtracking or fixture" > gpurun_out/r3_cfg2/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " gpurun_out/r3_cfg2/tests.txt | head
for rep in 1 2; do for fz in 1 0; do
  PMX_STEP_FUSED=$fz python bench.py --config cfg2 --steps 300 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$fz cfg2 it/s=%.1f ms/step=%.5f k1_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
PMX_STEP_FUSED=1 python bench.py --config cfg5 --steps 40 --warmup 10 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 fused it/s=%.1f' % d['value'])"
PMX_STEP_FUSED=0 python bench.py --config cfg5 --steps 40 --warmup 10 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 unfused it/s=%.1f' % d['value'])"
