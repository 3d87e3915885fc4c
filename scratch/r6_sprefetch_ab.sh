#!/bin/bash
# [r6] K1 <HH> producers: the NEXT slot's S fragments requested in front of the barrier that ends a slot (this build) against the read behind it (scratch/libpmx_base.so):
# bit-identity of the factors, then alternating bench lines on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
cat > /tmp/fix_hash.py <<'PY'
import sys, os, hashlib
from functools import partial
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import proxmin_amd as pm
import bench
ops = pm.operators
cases = ((2048, 4096, 64, dict(algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(ops.prox_unity_plus, axis=0), check_convergence=False), "adaprox-k64"),
         (16384, 16384, 64, dict(algorithm=pm.adaprox, scheme="amsgrad", prox_S=partial(ops.prox_unity_plus, axis=0), check_convergence=False), "cfg3"),
         (2048, 2048, 64, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2), "bsdmm-k64"),
         (1024, 1536, 64, {}, "pgm-k64"), (1000, 1500, 50, dict(algorithm=pm.adaprox, scheme="adam"), "ragged-k50"), (777, 1333, 64, dict(algorithm=pm.bsdmm), "ragged-bsdmm"))
for (M, N, K, kw, tag) in cases:
    Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Yd, A, S, max_iter=9, e_rel=1e-3, **kw)
    print(tag, M, N, K, hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12])
    del Yd
PY
python /tmp/fix_hash.py > $O/hash_new.txt 2>/dev/null
PMX_LIB=$PWD/scratch/libpmx_base.so python /tmp/fix_hash.py > $O/hash_base.txt 2>/dev/null
if diff -q $O/hash_new.txt $O/hash_base.txt > /dev/null; then echo "factors IDENTICAL ($(wc -l < $O/hash_new.txt) runs)"; cat $O/hash_new.txt; else echo "DIFFERENT:"; diff $O/hash_new.txt $O/hash_base.txt; fi
for rep in 1 2 3; do
for F in new base; do
  if [ $F = base ]; then export PMX_LIB=$PWD/scratch/libpmx_base.so; else unset PMX_LIB; fi
  python bench.py --skip-cpu-baseline --steps 20 --warmup 3 > $O/bench_${F}_$rep.json 2> /dev/null
  grep '^{' $O/bench_${F}_$rep.json | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
oc = d.get('other_configs', {})
def g(k, f): return oc.get(k, {}).get(f, float('nan'))
print('$F rep $rep | cfg3 %.1f it/s k1 %.4f tail %.4f | cfg4share %.1f k1 %.4f | cfg5 %.1f k1 %.4f tail %.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['tail_ms'],
      g('cfg4_share8192', 'value'), g('cfg4_share8192', 'k1_ms'), g('cfg5', 'value'), g('cfg5', 'k1_ms'), g('cfg5', 'tail_ms')))
"
done
done | tee $O/sprefetch_ab.txt
