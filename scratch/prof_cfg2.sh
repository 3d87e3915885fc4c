cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cfg2t
mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o s -- python bench.py --config cfg2 ${MODE:+--mode $MODE} --steps 200 --warmup 20 --no-cpu > $O/bench.json 2> $O/err.txt
python scratch/trace_gaps.py $(ls $O/t/*kernel_trace.csv) 20000 > $O/timeline.txt 2>/dev/null
sed -n 1,12p $O/timeline.txt
grep '^{' $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
find $O/t -name "*kernel_trace.csv" -delete
