#!/bin/bash
# [r6] the K = 128 counterpart of r6_chain_ablation.sh: k_grad_f16_k128<HH, RS, CHAIN> at cfg4's 8192-row share, K1 back to back on fixed factors.
# -DPMX_CHAIN_ABL bits: 1 no fetch / add of the previous sum, 2 no arrival look / wait, 4 arrival published without waiting for the stores, 8 gA flushed only once (no per-panel stores)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ad; mkdir -p $O; cd $R
for rep in 1 2; do
  for v in 0 1 2 4 8 15; do PMX_LIB=$R/scratch/libpmx_abl$v.so python scratch/r6_chain_ablation.py 8192x16384x128 2>&1 | grep "K1 back"; done
  PMX_K1_CHAIN=0 PMX_LIB=$R/scratch/libpmx_abl0.so python scratch/r6_chain_ablation.py 8192x16384x128 2>&1 | grep "K1 back"
done | tee $O/ablation_k128.txt
