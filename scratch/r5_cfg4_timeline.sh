#!/bin/bash
# kernel timeline of cfg4 whole on one GPU (65536 x 16384 x 128): what the 0.32 ms outside K1 consist of
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_j; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o s -- python bench.py --config cfg4 --steps 12 --warmup 5 --no-cpu > $O/cfg4.json 2> $O/cfg4.err
python scratch/trace_gaps.py $(ls $O/kt/*kernel_trace.csv | head -1) > $O/timeline_cfg4_1gpu.txt 2>&1
cp $(ls $O/kt/*kernel_stats.csv | head -1) $O/kernel_stats_cfg4_1gpu.csv
rm -rf $O/kt
tail -22 $O/timeline_cfg4_1gpu.txt
