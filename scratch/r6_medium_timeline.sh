#!/bin/bash
# [r6] kernel timeline of pgm at a medium K = 64 / 128 shape in the library default mode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
cat > /tmp/med.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K = [int(x) for x in sys.argv[1:4]]
Y, A0, S0 = bench.make_problem_device(M, N, K, False, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K)
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
run = bench.begin_solver(dev, sys.argv[4], False)
run(60)
torch.cuda.synchronize()
PY
for spec in "4096 4096 64 pgm" "4096 4096 128 pgm" "4096 4096 64 bsdmm"; do
  tag=$(echo $spec | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$tag -o s -- python /tmp/med.py $spec > /dev/null 2>&1
  echo "== $spec"; python scratch/trace_gaps.py $(ls $O/kt_$tag/*kernel_trace.csv | head -1) 20000 | tail -12
  rm -rf $O/kt_$tag
done
