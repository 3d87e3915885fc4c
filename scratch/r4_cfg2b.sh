#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_cfg2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_parity_strict.py tests/test_gpu_parity_long.py tests/test_gpu_callbacks.py -q -k "pgm or fista or fixture or medium or cfg2 or chained_iterations or convergence" > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests.txt | head
for rep in 1 2 3; do
for v in 0 1; do
  PMX_GRAM_IN_UPDATE=$v python bench.py --config cfg2 --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 gram_in_update=$v it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done
done 2>&1 | tee $O/ab.txt
python - <<'PY' 2>&1 | tee -a $O/ab.txt
# bit-identity: the fused partials give the very factors k_gram_partial's do at 4096 rows (same shares, same order)
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import nmf_oracle as orc
out = {}
for v in ("0", "1"):
    os.environ["PMX_GRAM_IN_UPDATE"] = v
    import proxmin_amd as pm
    for name, (M, N, K, acc) in {"4096x4096x32": (4096, 4096, 32, False), "4096x4096x64 fista": (4096, 4096, 64, True), "1000x3000x24": (1000, 3000, 24, False)}.items():
        Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=5)
        kw = dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)) if acc else {}
        pm.nmf.nmf(Y, A, S, max_iter=12, e_rel=1e-9, **kw)
        out[(name, v)] = hashlib.sha1(A.tobytes() + S.tobytes()).hexdigest()[:12]
for name in sorted({k[0] for k in out}):
    print(name, out[(name, "0")], out[(name, "1")], "IDENTICAL" if out[(name, "0")] == out[(name, "1")] else "differ (another grouping of the same sum)")
PY
