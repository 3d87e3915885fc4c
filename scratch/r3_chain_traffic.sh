#!/bin/bash
# round 3: K1's HBM traffic (FETCH x 2, WRITE; PMC, separate passes) against the chain length -- where do the writes beyond the
# final slabs come from?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain_traffic
mkdir -p $O
cd $R
for CH in 0 4 8 16 32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/$C
    PMX_K1_CHAIN=$CH rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- python bench.py --steps 8 --warmup 2 --no-cpu > /dev/null 2>&1
  done
  python - $CH $O <<'PY'
import csv, glob, os, sys
ch, O = sys.argv[1:3]
def avg(counter):
    path = glob.glob(os.path.join(O, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_grad" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    real = [x for x in v if x >= 0.05 * max(v)]
    return sum(real) / len(real), len(real)
f, nf = avg("FETCH_SIZE"); w, nw = avg("WRITE_SIZE")
print("PMX_K1_CHAIN=%-2s  FETCH x2 %.1f MB   WRITE %.1f MB   (launches %d / %d)" % (ch, 2 * f * 1024 / 1e6, w * 1024 / 1e6, nf, nw), flush=True)
PY
done | tee $O/summary.txt
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
