set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PMX_FORCE_SHARDED=1
for rows in 2048 4096; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard_$rows -o s -- python bench.py --rows $rows --steps 100 --warmup 20 --no-cpu > $O/shard_$rows.json 2> $O/shard_$rows.err
python scratch/trace_gaps.py $(ls $O/shard_$rows/*kernel_trace.csv) 20000 > $O/shard_${rows}_timeline.txt
find $O/shard_$rows -name "*kernel_trace.csv" -delete
tail -c 400 $O/shard_$rows.json
done
