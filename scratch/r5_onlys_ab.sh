#!/bin/bash
# cfg5 (bsdmm: one-gradient K1 passes): <ONLYS> (the S pass keeps its A fragments in registers) against the plain row split; PMX_K1_ROLE_SPLIT=0 switches both <RS> and <ONLYS> off
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gfix.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms') or 0))"; }
for i in 1 2 3; do
echo -n "ONLYS=1 "; python bench.py --config cfg5 --steps 60 --warmup 10 --no-cpu 2>/dev/null | line
echo -n "ONLYS=0 "; PMX_K1_ROLE_SPLIT=0 python bench.py --config cfg5 --steps 60 --warmup 10 --no-cpu 2>/dev/null | line
done
