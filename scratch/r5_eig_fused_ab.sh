#!/bin/bash
# A/B: k_gram_reduce + k_eig as one launch (k_gram_reduce_eig, default) against two (PMX_EIG_FUSED=0); cfg2 in three modes, cfg5
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do for F in 1 0; do
for spec in "cfg2 f16x2r 400 40" "cfg2 f32 400 40" "cfg5 f16x2r 40 10"; do
  set -- $spec
  PMX_EIG_FUSED=$F python bench.py --config $1 --mode $2 --steps $3 --warmup $4 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$F $1 $2 it/s %.1f ms %.4f k1 %.4f tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done; done; done
