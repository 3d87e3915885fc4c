#!/bin/bash
# upper-bound ablations of k_grad_f16_v8's three remaining levers (scratch/r4_v8_ablations.py builds the libraries)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_v8_abl; mkdir -p $O
for rep in 1 2 3; do
for v in base a b c abc; do
  [ $v = base ] && L=$R/proxmin_amd/libpmx.so || L=$R/scratch/libpmx_abl_$v.so
  TAG=$v PMX_LIB=$L timeout 120 python scratch/r4_k1_power.py 2>&1 | tail -1
done
done 2>&1 | tee $O/k1_b2b_power.txt
for rep in 1 2; do
for v in base a c; do
  [ $v = base ] && L=$R/proxmin_amd/libpmx.so || L=$R/scratch/libpmx_abl_$v.so
  PMX_LIB=$L timeout 200 python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bench it/s=%.1f ms/step=%.4f k1_ms=%.4f sub=%s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['sub_iterations_per_step']))"
done
done 2>&1 | tee $O/bench_ab.txt
