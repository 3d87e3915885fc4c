cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -o s -- python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu > $O/cfg5.json 2> $O/cfg5.err
python scratch/trace_gaps.py $(ls $O/cfg5/*kernel_trace.csv) 100000 > $O/cfg5_timeline.txt
find $O/cfg5 -name "*kernel_trace.csv" -delete
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2 -o s -- python bench.py --config cfg2 --mode f32 --steps 50 --warmup 10 --no-cpu > $O/cfg2.json 2> $O/cfg2.err
python scratch/trace_gaps.py $(ls $O/cfg2/*kernel_trace.csv) 20000 > $O/cfg2_timeline.txt
find $O/cfg2 -name "*kernel_trace.csv" -delete
