#!/bin/bash
# [r6] k_eig's wave solve with LDS broadcasts instead of v_readlane pairs (+ the fp64 matrix in LDS): bit-identity against the previous build (scratch/libpmx_base.so), then timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
cat > /tmp/eig_hash.py <<'PY'
import sys, os, hashlib
from functools import partial
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import proxmin_amd as pm
import bench
ops = pm.operators
cases = ((4096, 4096, 32, {}, "cfg2"), (1024, 1536, 64, {}, "pgm-k64"), (1000, 1500, 50, {}, "pgm-k50"), (200, 1000, 5, {}, "cfg1-f32"), (700, 900, 12, dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)), "fista-k12"),
         (2048, 2048, 64, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus, partial(ops.prox_soft, thresh=1e-3)]] * 2), "bsdmm-k64"),
         (1024, 2048, 128, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus], [ops.prox_plus]]), "bsdmm-k128"), (300, 400, 7, dict(algorithm=pm.bsdmm, proxs_g=[[ops.prox_plus], [ops.prox_plus]]), "bsdmm-k7"))
for (M, N, K, kw, tag) in cases:
    Yd, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
    A, S = A0.copy(), S0.copy()
    ret = pm.nmf.nmf(Yd, A, S, max_iter=12, e_rel=1e-6, **kw)
    steps = [float(x) for x in ret[2]] if isinstance(ret, tuple) and len(ret) == 3 else None
    print(tag, M, N, K, hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12], steps)
# fp64 small problem (k_small_f64 shares the wave solve)
from oracle import nmf_oracle as orc
Y, A0, S0 = orc.synthetic_problem(200, 1000, 5, np.float64, seed=3)
A, S = A0.copy(), S0.copy()
ret = pm.nmf.nmf(Y, A, S, max_iter=12, e_rel=1e-12)
print("f64", hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12], [float(x) for x in ret[2]])
PY
python /tmp/eig_hash.py > $O/hash_new.txt 2>/dev/null
PMX_LIB=$GRAFT_REPO_ROOT/scratch/libpmx_base.so python /tmp/eig_hash.py > $O/hash_base.txt 2>/dev/null
if diff -q $O/hash_new.txt $O/hash_base.txt > /dev/null; then echo "factors and steps IDENTICAL to the previous build ($(wc -l < $O/hash_new.txt) runs)"; else echo "DIFFERENT:"; diff $O/hash_new.txt $O/hash_base.txt; fi
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms') or 0))"; }
for i in 1 2 3; do
for L in new base; do
if [ $L = base ]; then export PMX_LIB=$GRAFT_REPO_ROOT/scratch/libpmx_base.so; else unset PMX_LIB; fi
echo -n "$L cfg2 f16x2r "; python bench.py --config cfg2 --mode f16x2r --steps 400 --warmup 40 --no-cpu 2>/dev/null | line
echo -n "$L cfg2 f32    "; python bench.py --config cfg2 --steps 400 --warmup 40 --no-cpu 2>/dev/null | line
echo -n "$L cfg5        "; python bench.py --config cfg5 --steps 40 --warmup 10 --no-cpu 2>/dev/null | line; echo "$L mediums:"; python scratch/r6_pgm_k64_probe.py 2>/dev/null | grep f16x2r
done
done | tee $O/eig_ab.txt
unset PMX_LIB
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f64.py -m gpu -x -q 2>&1 | tail -3
