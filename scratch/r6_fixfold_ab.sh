#!/bin/bash
# [r6] mode f16x2r: k_gfix_gram BEFORE K1, the fold of its partials in K1's first workgroups, k_gfix_apply behind K1 (PMX_FIX_FOLD_IN_K1=1, default) against the three
# launches behind K1 (= 0): bit-identity of the factors, then alternating bench lines (cfg3 headline + side configurations) on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
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
         (2048, 2048, 128, dict(algorithm=pm.adaprox, scheme="adam"), "adaprox-k128"),
         (8192, 16384, 128, dict(algorithm=pm.adaprox, scheme="amsgrad", check_convergence=False), "cfg4-share"),
         (2048, 2048, 64, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2), "bsdmm-k64"),
         (1024, 1536, 64, {}, "pgm-k64"), (1000, 1500, 50, dict(algorithm=pm.adaprox, scheme="adam"), "ragged-k50"))
for (M, N, K, kw, tag) in cases:
    Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    A, S = A0.copy(), S0.copy()
    pm.nmf.nmf(Yd, A, S, max_iter=9, e_rel=1e-3, **kw)
    print(tag, M, N, K, hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12])
    del Yd
PY
PMX_FIX_FOLD_IN_K1=1 python /tmp/fix_hash.py > $O/hash_1.txt 2>/dev/null
PMX_FIX_FOLD_IN_K1=0 python /tmp/fix_hash.py > $O/hash_0.txt 2>/dev/null
if diff -q $O/hash_1.txt $O/hash_0.txt > /dev/null; then echo "factors IDENTICAL with and without the correction's fold in K1 ($(wc -l < $O/hash_1.txt) runs)"; cat $O/hash_1.txt; else echo "DIFFERENT:"; diff $O/hash_1.txt $O/hash_0.txt; fi
for rep in 1 2 3; do
for F in 1 0; do
  PMX_FIX_FOLD_IN_K1=$F python bench.py --skip-cpu-baseline > $O/bench_fold${F}_$rep.json 2> /dev/null
  grep '^{' $O/bench_fold${F}_$rep.json | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
oc = d.get('other_configs', {})
def g(k, f): return oc.get(k, {}).get(f, float('nan'))
print('fixfold $F rep $rep | cfg3 %.1f it/s k1 %.4f tail %.4f | cfg4share %.1f k1 %.4f tail %.4f | cfg5 %.1f k1 %.4f tail %.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['tail_ms'],
      g('cfg4_share8192', 'value'), g('cfg4_share8192', 'k1_ms'), g('cfg4_share8192', 'tail_ms'), g('cfg5', 'value'), g('cfg5', 'k1_ms'), g('cfg5', 'tail_ms')))
"
done
done | tee $O/fixfold_ab.txt
timeout 900 python -m pytest tests/test_gpu_gfix.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
