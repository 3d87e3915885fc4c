#!/bin/bash
# K1 iteration loop: kernel parity tests + default bench (no CPU leg) + back-to-back K1 timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_k1
O=gpurun_out/r3_k1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s=%.1f ms/step=%.4f k1_ms=%.4f frac=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; done | tee $O/bench.txt
