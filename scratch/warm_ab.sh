#!/bin/bash
# does the K1 time of a 20-step timed region depend on how long the GPU has been busy before it?
for W in 5 40 100 300; do
  python bench.py --steps 20 --warmup $W --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup=$W eff=%d it/s=%.1f ms/step=%.4f k1_ms=%.4f sub=%s' % (d['warmup_effective'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['sub_iterations_per_step']))"
done
