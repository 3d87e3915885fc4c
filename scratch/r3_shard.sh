#!/bin/bash
# one rank's share of an 8-GPU run on one GPU (local collectives): cfg4 (8192 rows, K=128) with and without the S-split, cfg3 (2048 rows)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_shard
O=gpurun_out/r3_shard
export PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ms/step=%.4f k1_ms=%.4f it/s=%.1f %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'], d['config']['parallelism'][:90]))"; }
PMX_S_SPLIT=1 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu 2>/dev/null | tee $O/cfg4_share_split.json | p cfg4-split
PMX_S_SPLIT=0 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu 2>/dev/null | tee $O/cfg4_share_repl.json | p cfg4-replicated
python bench.py --config cfg3 --rows 2048 --steps 60 --warmup 20 --no-cpu 2>/dev/null | tee $O/cfg3_share.json | p cfg3-share
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMX_S_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_split -o p -- python $R/bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > /dev/null 2>&1
cd $R; python scratch/trace_gaps.py $(find $O/prof_split -name "*kernel_trace.csv" | head -1) 2>/dev/null | tail -25 > $O/timeline_cfg4_share_split.txt; cat $O/timeline_cfg4_share_split.txt | tail -14
find $O/prof_split -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_cfg4_share_split.csv \;; rm -rf $O/prof_split
