#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_cfg2_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
for m in f32 f16x2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python bench.py --config cfg2 --mode $m --steps 100 --warmup 20 --no-cpu > $O/trace_$m.json 2> $O/trace.err
python scratch/trace_gaps.py $(ls $O/trace/*kernel_trace.csv) 15000 > $O/timeline_cfg2_$m.txt
rm -rf $O/trace
echo "== $m"; head -9 $O/timeline_cfg2_$m.txt
done
