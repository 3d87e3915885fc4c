#!/bin/bash
# [r6] pgm: the step rule's Gram fold riding in K1's first workgroups (PMX_FOLD_IN_K1=1, the default) against the k_gram_reduce launch (= 0): bit-identity of
# the factors, then alternating bench lines of cfg2 in both arithmetic modes on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
cat > /tmp/fold_hash.py <<'PY'
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import proxmin_amd as pm
import bench
for (M, N, K, kw, tag) in ((4096, 4096, 32, {}, "pgm"), (4096, 4096, 32, dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)), "fista"), (1024, 2048, 32, {}, "pgm-small"), (2048, 1024, 24, {}, "pgm-K24")):
    for mode in ("f32", "f16x2r", "f16x2"):
        pm.set_default_mode(mode)
        Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
        A, S = A0.copy(), S0.copy()
        ret = pm.nmf.nmf(Yd, A, S, max_iter=23, e_rel=1e-4, **kw)      # (e_rel 1e-4: the stopping test may fire inside the run)
        print(tag, M, N, K, mode, hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12], ret[0], [float(x) for x in ret[2]])
PY
PMX_FOLD_IN_K1=1 python /tmp/fold_hash.py > $O/hash_fold1.txt 2>/dev/null
PMX_FOLD_IN_K1=0 python /tmp/fold_hash.py > $O/hash_fold0.txt 2>/dev/null
if diff -q $O/hash_fold1.txt $O/hash_fold0.txt > /dev/null; then echo "factors, verdicts and steps IDENTICAL with and without the fold in K1 ($(wc -l < $O/hash_fold1.txt) runs)"; else echo "DIFFERENT:"; diff $O/hash_fold1.txt $O/hash_fold0.txt; fi
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms') or 0))"; }
for i in 1 2 3; do
for F in 1 0; do
echo -n "fold=$F cfg2 f16x2r "; PMX_FOLD_IN_K1=$F python bench.py --config cfg2 --mode f16x2r --steps 400 --warmup 40 --no-cpu 2>/dev/null | line
echo -n "fold=$F cfg2 f32    "; PMX_FOLD_IN_K1=$F python bench.py --config cfg2 --steps 400 --warmup 40 --no-cpu 2>/dev/null | line
done
done | tee $O/fold_ab.txt
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_kernels.py tests/test_gpu_callbacks.py -m gpu -x -q -k "pgm or fista or fixture or step or kernel" 2>&1 | tail -3
