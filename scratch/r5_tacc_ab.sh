#!/bin/bash
# <RS> gSt waves: the panel's contribution in a fresh accumulator + IEEE add (parity against the yardstick at full cfg3, then speed)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity_long.py -m gpu -x -q -k "test_cfg3_full_size_device_error" 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_long.json'))
for k,v in d.items():
    if 'cfg3 full, 3 its' in k: print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a.startswith('out') or a=='worst_ratio'})
PY
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2; do python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | line; done
