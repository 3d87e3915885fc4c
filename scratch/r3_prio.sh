#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do for p in 0 1 2; do
  PMX_K1_PRIO=$p python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$p it/s=%.1f ms/step=%.4f k1_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
