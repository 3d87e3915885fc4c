#!/bin/bash
# A/B of a K1 change: kernel tests of the touched kernels, then cfg3 / cfg2 bench lines (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gfix.py -m gpu -x -q 2>&1 | tail -3
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s | it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['dtype'][:7], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('tail_ms') or 0))"; }
for i in 1 2; do
python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | tee $O/bench_cfg3_$i.json | line
python bench.py --config cfg2 --mode f16x2r --steps 400 --warmup 40 --no-cpu 2>/dev/null | tee $O/bench_cfg2_f16x2r_$i.json | line
python bench.py --config cfg2 --steps 400 --warmup 40 --no-cpu 2>/dev/null | tee $O/bench_cfg2_f32_$i.json | line
done
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tee $O/bench_20_5.json | line
