#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_shard; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_gpu_distributed_world2.py tests/test_gpu_distributed.py -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for cfg in "cfg4 8192" "cfg3 2048"; do set -- $cfg
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 timeout 300 python bench.py --config $1 --rows $2 --steps 40 --warmup 10 --no-cpu > $O/bench_$1_rows$2_rank0of8.json 2> $O/bench_$1.err
python -c "import json; d=json.load(open('$O/bench_$1_rows$2_rank0of8.json')); print('$1 rows $2', d['value'], d['ms_per_step'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phases_ms'].items()})"
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o s -- python bench.py --config $1 --rows $2 --steps 40 --warmup 10 --no-cpu > $O/trace.json 2> $O/trace.err
python scratch/trace_gaps.py $(ls $O/trace/*kernel_trace.csv) 20000 > $O/timeline_$1_rows$2.txt
rm -rf $O/trace
head -12 $O/timeline_$1_rows$2.txt
done
rocprofv3 -L > $O/rocprof_counters.txt 2>&1; grep -i -c "TCC" $O/rocprof_counters.txt; grep -o "TCC_[A-Z0-9_]*WRITEBACK[A-Za-z0-9_]*\|TCC_[A-Z0-9_]*EVICT[A-Za-z0-9_]*\|TCC_EA0*_WRREQ[A-Za-z0-9_]*" $O/rocprof_counters.txt | sort -u | head -40
