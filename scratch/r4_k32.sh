#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_k32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "k32" > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests.txt | head -20
for v in f32 f16x2; do
  timeout 200 python bench.py --config cfg2 --mode $v --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 $v it/s=%.1f ms/step=%.4f k1=%s k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done 2>&1 | tee $O/bench.txt
