#!/bin/bash
# round 3, first GPU session: baseline tests + bench + power/clock artefact + Y streaming tables
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_first
O=gpurun_out/r3_first
( timeout 120 ./scratch/ystream2 > $O/ystream2.txt 2>&1 ) 
( timeout 120 ./scratch/ystream > $O/ystream_zero_r2.txt 2>&1 )
timeout 300 python scratch/r3_power_probe.py > $O/power_probe.txt 2>&1
cp -r gpurun_out/r3_power $O/ 2>/dev/null
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cat $O/power_probe.txt | tail -20
tail -1 $O/bench_default.json | cut -c1-600
