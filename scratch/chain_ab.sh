#!/bin/bash
# same-box A/B of the chained gA accumulation: K1 average launch time and iterations/s per chain length
mkdir -p gpurun_out/chain_ab
for L in 0 32 16 8 4 0 32; do
  PMX_K1_CHAIN=$L python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain=$L it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_us=%.1f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], 1e3*(d['ms_per_step']-d['roofline']['avg_launch_ms'])))"
done | tee gpurun_out/chain_ab/result.txt
