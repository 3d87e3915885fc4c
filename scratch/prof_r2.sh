#!/bin/bash
# round-2 profile of the default bench: kernel stats + per-iteration timeline (+ PMC traffic when PMC=1)
TAG=${1:-r02_a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o s -- python bench.py --steps 100 --warmup 20 --no-cpu > $O/bench_under_rocprof.json 2> $O/kt.err
python scratch/trace_gaps.py $(ls $O/kt/*kernel_trace.csv | head -1) > $O/timeline.txt 2>&1
cp $(ls $O/kt/*kernel_stats.csv | head -1) $O/kernel_stats.csv
if [ "$PMC" = "1" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o f -- python bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o f -- python bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  python profiles/summarize_pmc.py $O/pmc_summary.json fetch=$(ls $O/fetch/*counter_collection.csv) write=$(ls $O/write/*counter_collection.csv) sq=$(ls $O/sq/*counter_collection.csv)
  rm -rf $O/fetch $O/write $O/sq
fi
rm -rf $O/kt
tail -24 $O/timeline.txt
head -12 $O/kernel_stats.csv | cut -c1-160
