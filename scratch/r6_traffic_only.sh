#!/bin/bash
# re-measure profiles/k1_traffic.json for the current build (every config the bench quotes), then the driver-flag and default bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_y; mkdir -p $O
./scratch/measure_traffic.sh cfg3 f16x2r > $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg3 f16x2 >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg3 f32 >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg5 f16x2r >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg2 f32 >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg2 f16x2r >> $O/traffic.log 2>&1
./scratch/measure_traffic.sh cfg4 f16x2r 8192 >> $O/traffic.log 2>&1
cp gpurun_out/k1_traffic.json profiles/k1_traffic.json
cp gpurun_out/k1_traffic.json $O/k1_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for f in bench_20_5 bench_default; do grep '^{' $O/$f.json | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$f', round(d['value'], 1), d['ms_per_step'], 'k1', d['roofline']['avg_launch_ms'], 'frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline'].get('traffic'), 'floor', d['roofline'].get('frac_of_energy_floor'))
print('   fp64', d.get('fp64_inputs', {}).get('value'), d.get('fp64_inputs', {}).get('frac_of_measured_fp64_mfma_peak'), 'e2e', d.get('end_to_end', {}).get('end_to_end_ms'), 'cpu', d.get('cpu_baseline', {}).get('value'))
"; done
