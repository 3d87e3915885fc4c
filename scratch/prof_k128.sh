# cfg4 (K = 128) on one GPU: the whole problem and one rank's 8192-row share (sharded code path, world 1), both arithmetics,
# plus the kernel timeline of the share in mode f16x2
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/k128
mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for mode in f16x2 f32; do
timeout 300 python bench.py --config cfg4 --mode $mode --steps 20 --warmup 5 --no-cpu > $O/bench_cfg4_1gpu_$mode.json 2> $O/bench_cfg4_1gpu_$mode.err
PMX_FORCE_SHARDED=1 timeout 300 python bench.py --config cfg4 --mode $mode --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg4_shard8192_$mode.json 2> $O/bench_cfg4_shard8192_$mode.err
done
PMX_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/trace.json 2> $O/trace.err
python scratch/trace_gaps.py $(ls $O/trace/*kernel_trace.csv) 20000 > $O/timeline_cfg4_shard8192_f16x2.txt
cp $(ls $O/trace/*kernel_stats.csv) $O/kernel_stats_cfg4_shard8192_f16x2.csv
find $O/trace -name "*kernel_trace.csv" -delete
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
head -30 $O/timeline_cfg4_shard8192_f16x2.txt
