#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6f64; mkdir -p $O
CFGS=${CFGS:-"cfg3 cfg4share"}
for c in $CFGS; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o s -- python scratch/r6_f64_bench.py $c > $O/bench_$c.txt 2> $O/kt_$c.err
cat $O/bench_$c.txt
f=$(find $O/kt_$c -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_${c}_f64.csv; rm -rf $O/kt_$c
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_${c}_f64.csv")))
for r in rows[:14]:
    print("%-60s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
