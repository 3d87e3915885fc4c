#!/bin/bash
# [r6] how much of K1 is the gA waves' per-panel flush?  -DPMX_CHAIN_ABL=8: the tile is stored only after the workgroup's last panel (timing only: WRONG gradients)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ad; mkdir -p $O; cd $R
for rep in 1 2; do
  for v in 0 8; do PMX_LIB=$R/scratch/libpmx_abl$v.so python scratch/r6_chain_ablation.py 2>&1 | grep "K1 back"; done
done | tee $O/flush_k64.txt
