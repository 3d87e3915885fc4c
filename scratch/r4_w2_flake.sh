#!/bin/bash
# hunt the rare world-2 hang: the one case that timed out once in a full run, many times, short patience
cd $GRAFT_REPO_ROOT
export PMX_W2_TIMEOUT=60
for i in $(seq 1 40); do
  timeout 400 python -m pytest tests/test_gpu_distributed_world2.py -q -x -k "adaprox_k64_split or adaprox_k64_blocks or adaprox_unity" > gpurun_out/w2_hunt_$i.txt 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/w2_hunt_$i.txt)"
  if [ $rc -ne 0 ]; then grep -n "stuck\|rank \|File \|did not finish" gpurun_out/w2_hunt_$i.txt | head -60; break; fi
  rm -f gpurun_out/w2_hunt_$i.txt
done
