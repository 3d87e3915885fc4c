#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_second
O=gpurun_out/r3_second
( timeout 200 ./scratch/ystream2 > $O/ystream2.txt 2>&1 )
timeout 300 python scratch/r3_power_probe.py > $O/power_probe.txt 2>&1
cp -r gpurun_out/r3_power $O/ 2>/dev/null
cat $O/power_probe.txt | tail -20
head -40 $O/ystream2.txt
