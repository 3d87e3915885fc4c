#!/bin/bash
# [r6] the step rule's eigen-solve on a side stream, behind K1 and beside the correction's launches (PMX_SIDE_EIG=1, default) against the single stream (= 0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
cat > /tmp/side_hash.py <<'PY'
import sys, os, hashlib
from functools import partial
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import proxmin_amd as pm
import bench
ops = pm.operators
cases = ((4096, 4096, 32, {}, "cfg2"), (4096, 4096, 64, {}, "pgm-k64"), (2048, 4096, 128, {}, "pgm-k128"), (1000, 1500, 50, {}, "pgm-k50"), (200, 1000, 5, {}, "cfg1-f32"),
         (700, 900, 12, dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)), "fista-k12"), (4096, 4096, 64, dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)), "fista-k64"),
         (2048, 2048, 64, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2), "bsdmm-k64"),
         (1024, 2048, 128, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus], [ops.prox_plus]]), "bsdmm-k128"), (300, 400, 7, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus], [ops.prox_plus]]), "bsdmm-k7"))
for mode in ("f16x2r", "f32"):
    pm.set_default_mode(mode)
    for (M, N, K, kw, tag) in cases:
        Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
        for e_rel, its in ((1e-9, 40), (2e-2, 300)):
            A, S = A0.copy(), S0.copy()
            ret = pm.nmf.nmf(Yd, A, S, max_iter=its, e_rel=e_rel, **kw)
            steps = [float(x) for x in ret[2]] if isinstance(ret, tuple) and len(ret) == 3 else ret
            print(mode, tag, M, N, K, e_rel, hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12], steps)
PY
PMX_SIDE_EIG=1 python /tmp/side_hash.py > $O/hash_1.txt 2>/dev/null
PMX_SIDE_EIG=0 python /tmp/side_hash.py > $O/hash_0.txt 2>/dev/null
if diff -q $O/hash_1.txt $O/hash_0.txt > /dev/null; then echo "factors, steps and verdicts IDENTICAL with and without the side stream ($(wc -l < $O/hash_1.txt) runs)"; else echo "DIFFERENT:"; diff $O/hash_1.txt $O/hash_0.txt | head; fi
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms') or 0))"; }
for i in 1 2 3; do
for F in 1 0; do
export PMX_SIDE_EIG=$F
echo -n "side=$F cfg2 f16x2r "; python bench.py --config cfg2 --mode f16x2r --steps 400 --warmup 40 --no-cpu 2>/dev/null | line
echo -n "side=$F cfg2 f32    "; python bench.py --config cfg2 --steps 400 --warmup 40 --no-cpu 2>/dev/null | line
echo -n "side=$F cfg5        "; python bench.py --config cfg5 --steps 40 --warmup 10 --no-cpu 2>/dev/null | line
echo "side=$F mediums:"; python scratch/r6_pgm_k64_probe.py 2>/dev/null
done
done | tee $O/side_eig_ab.txt
unset PMX_SIDE_EIG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nmf.py -m gpu -x -q 2>&1 | tail -3
