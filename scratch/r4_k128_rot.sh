#!/bin/bash
# K = 128 K1: chain / slabs / slabs with the chain's panel rotation (ablation library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for rep in 1 2; do
echo "== chain (default)";            python scratch/k128_time.py 2>&1 | grep -E "doA|k1"
echo "== slabs (PMX_K1_CHAIN=0)";     PMX_K1_CHAIN=0 python scratch/k128_time.py 2>&1 | grep -E "doA|k1"
echo "== slabs + rotation (ablation)"; PMX_K1_CHAIN=0 PMX_LIB=$PWD/scratch/libpmx_abl_rot.so python scratch/k128_time.py 2>&1 | grep -E "doA|k1"
done
} > gpurun_out/r4_k128_rot.txt 2>&1
cat gpurun_out/r4_k128_rot.txt
