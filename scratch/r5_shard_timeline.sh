#!/bin/bash
# kernel timeline of one rank's share of cfg4 (8192 x 16384 x 128): sharded code path (one process playing rank 0 of 8) beside the single-GPU path at the same shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_f; mkdir -p $O
cd $R
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sh -o s -- python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/shard.json 2> $O/shard.err
python scratch/trace_gaps.py $(ls $O/kt_sh/*kernel_trace.csv | head -1) > $O/timeline_cfg4_shard8192.txt 2>&1
cp $(ls $O/kt_sh/*kernel_stats.csv | head -1) $O/kernel_stats_cfg4_shard8192.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_1 -o s -- python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/single.json 2> $O/single.err
python scratch/trace_gaps.py $(ls $O/kt_1/*kernel_trace.csv | head -1) > $O/timeline_cfg4_rows8192_single.txt 2>&1
cp $(ls $O/kt_1/*kernel_stats.csv | head -1) $O/kernel_stats_cfg4_rows8192_single.csv
rm -rf $O/kt_sh $O/kt_1
tail -30 $O/timeline_cfg4_shard8192.txt
echo ======
tail -24 $O/timeline_cfg4_rows8192_single.txt
