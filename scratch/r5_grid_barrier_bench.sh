#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_g
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gb scratch/r5_grid_barrier_bench.hip 2>/dev/null
timeout 120 /tmp/gb | tee gpurun_out/r05_g/grid_barrier.txt
