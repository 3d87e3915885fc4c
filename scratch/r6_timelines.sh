#!/bin/bash
# [r6] per-iteration timelines (rocprofv3 --kernel-trace) of every configuration the bench line quotes + K1's time along one run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
tl() {  # tag, bench flags
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1 -o s -- python bench.py $2 --no-cpu > $O/bench_under_rocprof_$1.json 2> /dev/null
  python scratch/trace_gaps.py $(ls $O/kt_$1/*kernel_trace.csv | head -1) > $O/timeline_$1.txt 2>&1
  cp $(ls $O/kt_$1/*kernel_stats.csv | head -1) $O/kernel_stats_$1.csv
  rm -rf $O/kt_$1
  echo "== $1"; tail -14 $O/timeline_$1.txt
}
tl cfg3 "--steps 100 --warmup 20"
tl cfg5 "--config cfg5 --steps 30 --warmup 10"
tl cfg2r "--config cfg2 --mode f16x2r --steps 200 --warmup 40"
tl cfg2 "--config cfg2 --steps 200 --warmup 40"
tl cfg4rows8192 "--config cfg4 --rows 8192 --steps 40 --warmup 20"
python scratch/r6_k1_vs_iteration.py 2>/dev/null | tee $O/k1_vs_iteration.txt
