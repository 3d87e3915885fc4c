#!/bin/bash
# [r6] A/B: s_setprio level of K1's consumer waves (PMX_K1_PRIO; < 0: the producers instead), same box, alternating.
# One bench.py process per setting (cfg3 headline at the default flags + the side configurations of the same line).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
for rep in 1 2; do
for P in 0 1 2 3 -1; do
  PMX_K1_PRIO=$P python bench.py --skip-cpu-baseline > $O/bench_prio${P}_$rep.json 2> /dev/null
  grep '^{' $O/bench_prio${P}_$rep.json | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
oc = d.get('other_configs', {})
def g(k, f): return oc.get(k, {}).get(f, float('nan'))
print('prio %3s rep $rep | cfg3 %.1f it/s k1 %.4f tail %.4f | cfg4share %.1f k1 %.4f | cfg5 %.1f k1 %.4f | cfg2r %.0f k1 %.4f | cfg2 f32 %.0f | f16x2 %.1f' % ('$P', d['value'], d['roofline']['avg_launch_ms'], d['tail_ms'],
      g('cfg4_share8192', 'value'), g('cfg4_share8192', 'k1_ms'), g('cfg5', 'value'), g('cfg5', 'k1_ms'), g('cfg2_f16x2r', 'value'), g('cfg2_f16x2r', 'k1_ms'), g('cfg2', 'value'), d.get('value_f16x2_mode', {}).get('value', float('nan'))))
"
done
done | tee $O/prio_ab.txt
