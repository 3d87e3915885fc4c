#!/bin/bash
# [r6] fp64 at size: the numbers DESIGN 4.4 quotes (tag r06_p): bench of the three back-ends, kernel stats of cfg3 / cfg4's share under rocprofv3, the stand-alone pass
# harness, the measured fp64 MFMA rate, the randomized sweep (weights included)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_p; mkdir -p $O
python scratch/r6_f64_peak.py > $O/fp64_mfma_peak.txt 2>&1; cat $O/fp64_mfma_peak.txt
for s in "16384 16384 64" "8192 16384 128" "4096 4096 32" "8192 8192 64" "5000 7001 50"; do ./scratch/f64pass_bench $s 5; done > $O/f64_pass_harness.txt 2>&1; cat $O/f64_pass_harness.txt
python scratch/r6_f64_bench.py cfg2 cfg3 cfg4share cfg5 > $O/f64_bench.txt 2>&1; cat $O/f64_bench.txt
for c in cfg3 cfg4share; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o s -- python scratch/r6_f64_bench.py $c > /dev/null 2> $O/kt_$c.err
f=$(find $O/kt_$c -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_${c}_f64.csv; rm -rf $O/kt_$c $O/kt_$c.err
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_${c}_f64.csv")))
print("== $c")
for r in rows[:16]:
    print("%-60s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done | tee $O/kernel_stats_summary.txt
timeout 1500 python scratch/r6_fuzz_f64big.py 93 94 95 2>&1 | grep -v "RuntimeWarning\|return (P\|amdgpu.ids" | tee $O/fuzz_f64_at_size.txt
