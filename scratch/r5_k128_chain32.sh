#!/bin/bash
# K = 128: does a chain of 32 (ONE chain per XCD: 2 MB of live hand-off tiles instead of 4 MB = the XCD's whole L2) keep the hand-off in L2?
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for L in 16 32; do
  PMX_K1_CHAIN=$L python bench.py --config cfg4 --rows 8192 --steps 60 --warmup 20 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain $L: it/s %.1f ms %.4f k1 %.4f tail %.4f slabs %s chain %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms'], d['roofline']['k1_layout']['slabs_A'], d['roofline']['k1_layout']['chain']))"
done
done
for L in 16 32; do
  export PMX_K1_CHAIN=$L
  ./scratch/measure_traffic.sh cfg4 f16x2r 8192 > /dev/null 2>&1
  python -c "import json; d=json.load(open('gpurun_out/k1_traffic.json'))['cfg4_rows8192/f16x2r']; print('chain $L: fetch %.1f MB write %.1f MB' % (d['fetch_bytes']/1e6, d['write_bytes']/1e6))"
done
rm -f gpurun_out/k1_traffic.json
