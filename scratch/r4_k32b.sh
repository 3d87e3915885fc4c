#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_k32; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests_all.txt | head -20
for rep in 1 2; do
for v in f32 f16x2; do
  timeout 200 python bench.py --config cfg2 --mode $v --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 $v it/s=%.1f ms/step=%.4f k1=%s k1_ms=%.4f tail_ms=%.4f frac=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['tail_ms'], d['roofline']['frac']))"
done
PMX_K1_K32=0 timeout 200 python bench.py --config cfg2 --mode f16x2 --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 f16x2 with PMX_K1_K32=0 (rounds 1-3 path) it/s=%.1f k1=%s k1_ms=%.4f' % (d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms']))"
done 2>&1 | tee $O/bench.txt
