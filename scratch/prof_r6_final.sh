#!/bin/bash
# round-6 measurement batch (tag = $1, default r06_z): smoke, PMC traffic of this build for every config the bench quotes, kernel stats + timeline + SQ counters of the
# default bench, the bench lines (driver flags, default flags, other modes / configurations, one rank's share of the sharded runs).  The GPU test suite: its own call.
TAG=${1:-r06_z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
./scratch/measure_traffic.sh cfg3 f16x2r > $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg3 f16x2 >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg3 f32 >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg5 f16x2r >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg2 f32 >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg2 f16x2r >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg4 f16x2r 8192 >> $O/traffic.log 2>&1
cp gpurun_out/k1_traffic.json profiles/k1_traffic.json
cp gpurun_out/k1_traffic.json $O/k1_traffic.json
PMC=1 ./scratch/prof_r2.sh $TAG > $O/prof.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode f32 --no-cpu > $O/bench_f32.json 2>/dev/null
python bench.py --mode f16x2 --no-cpu > $O/bench_f16x2.json 2>/dev/null
python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu > $O/bench_cfg2.json 2>/dev/null
python bench.py --config cfg2 --mode f16x2r --steps 200 --warmup 20 --no-cpu > $O/bench_cfg2_f16x2r.json 2>/dev/null
python bench.py --config cfg5 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg5.json 2>/dev/null
python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg4_1gpu.json 2>/dev/null
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --rows 2048 --steps 100 --warmup 20 --no-cpu > $O/bench_shard2048_rank0of8.json 2>/dev/null
PMX_FORCE_SHARDED=1 PMX_BENCH_FAKE_WORLD=8 python bench.py --config cfg4 --rows 8192 --steps 40 --warmup 10 --no-cpu > $O/bench_cfg4_shard8192_rank0of8.json 2>/dev/null
for f in $O/bench_*.json; do echo "== $f"; grep '^{' $f | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], '|', d['dtype'][:8], '| it/s %.1f | ms %.4f | k1 %.4f | frac %.3f | traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('traffic')))"; done
tail -30 $O/timeline.txt
python scratch/r6_k1_vs_iteration.py 2>/dev/null > $O/k1_vs_iteration.txt
python scratch/r5_power_probe.py > $O/power.txt 2>&1 || true
