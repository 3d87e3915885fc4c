#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_cfg2; mkdir -p $O
for rep in 1 2 3; do
for v in 0 1; do
  PMX_SIDE_STEPS=$v python bench.py --config cfg2 --steps 400 --warmup 50 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 side_steps=$v it/s=%.1f ms/step=%.4f k1_ms=%.4f tail_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"
done
done 2>&1 | tee $O/ab_side_steps.txt
for v in 0 1; do
  PMX_SIDE_STEPS=$v python bench.py --config cfg5 --steps 30 --warmup 10 --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 side_steps=$v it/s=%.1f ms/step=%.4f k1_ms=%.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done 2>&1 | tee -a $O/ab_side_steps.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python bench.py --config cfg2 --steps 100 --warmup 20 --no-cpu > $O/trace.json 2> $O/trace.err
python scratch/trace_gaps.py $(ls $O/trace/*kernel_trace.csv) 20000 > $O/timeline_cfg2_side.txt
rm -rf $O/trace
cat $O/timeline_cfg2_side.txt | head -30
timeout 600 python -m pytest tests/test_gpu_nmf.py tests/test_gpu_parity_strict.py -q -x -k "pgm or fista or fixture or medium" 2>&1 | tail -3
