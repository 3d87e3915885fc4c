#!/bin/bash
# cfg3: chain length of k_grad_f16_v8<HH> against iteration / K1 / tail time
cd $GRAFT_REPO_ROOT
for L in 16 32 8 4 0; do
  PMX_K1_CHAIN=$L python bench.py --config cfg3 --steps 60 --warmup 20 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain $L: it/s %.1f ms %.4f k1 %.4f tail %.4f slabs %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms'], d['roofline']['k1_layout']['slabs_A']))"
done
