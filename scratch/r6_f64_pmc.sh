#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6f64; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $O/pmc -o p -- ./scratch/f64pass_bench ${SHAPE:-16384 16384 64} 2 > $O/pmc.out 2> $O/pmc.err
f=$(find $O/pmc -name '*counter_collection.csv' | head -1)
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "grad_pass" not in k: continue
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
rm -rf $O/pmc
