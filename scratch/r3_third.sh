#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_third
O=gpurun_out/r3_third
timeout 300 python scratch/r3_ablate.py > $O/ablate.txt 2>&1
cat $O/ablate.txt | grep -v "^$" | tail -40
