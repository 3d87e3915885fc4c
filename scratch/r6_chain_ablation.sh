#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_r; mkdir -p $O; cd $R
for rep in 1 2; do
  for v in 0 1 2 4 7; do PMX_LIB=$R/scratch/libpmx_abl$v.so python scratch/r6_chain_ablation.py 2>&1 | grep "K1 back"; done
  PMX_K1_CHAIN=0 PMX_LIB=$R/scratch/libpmx_abl0.so python scratch/r6_chain_ablation.py 2>&1 | grep "K1 back"
done | tee $O/ablation_k64.txt
