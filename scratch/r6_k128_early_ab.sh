#!/bin/bash
# [r6] K = 128: the gA hand-off one slot earlier (libpmx_early1.so: arrival looked at in the first slot of a panel, previous sum fetched and added in the second and third)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ag; mkdir -p $O; cd $R
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l)
print('it/s %.1f | ms %.4f | k1 %.4f | tail %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['tail_ms']))"; }
for rep in 1 2 3; do
  for v in early0 early1; do
    echo -n "rep $rep $v cfg4 share : "; PMX_LIB=$R/scratch/libpmx_$v.so python bench.py --config cfg4 --rows 8192 --no-cpu --steps 40 --warmup 20 2>/dev/null | line
  done
done | tee $O/ab.txt
for v in early0 early1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc
    PMX_LIB=$R/scratch/libpmx_$v.so rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc -o p -- python bench.py --config cfg4 --rows 8192 --mode f16x2r --steps 8 --warmup 2 --no-cpu > /dev/null 2>&1
    python - $v $C $O <<'PY'
import csv, glob, os, sys
v, C, O = sys.argv[1:4]
path = glob.glob(os.path.join(O, "pmc", "**", "*counter_collection.csv"), recursive=True)[0]
x = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_grad" in r["Kernel_Name"] and r["Counter_Name"] == C]
real = [y for y in x if y >= 0.05 * max(x)]
print("%s %s per K1 launch: %.1f MB (%d launches; KiB -> bytes%s)" % (v, C, sum(real) / len(real) * 1024 * (2 if C == "FETCH_SIZE" else 1) / 1e6, len(real), ", x 2" if C == "FETCH_SIZE" else ""))
PY
    rm -rf $O/pmc
  done
done | tee $O/traffic.txt
for v in early0 early1; do echo "== $v"; PMX_LIB=$R/scratch/libpmx_$v.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -1; PMX_LIB=$R/scratch/libpmx_$v.so python scratch/r6_chain_ablation.py 8192x16384x128 2>&1 | grep "K1 back"; done | tee $O/correctness_b2b.txt
