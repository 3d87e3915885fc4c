#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6f64; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS"; do
rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc -o p -- ./scratch/f64pass_bench ${SHAPE:-4096 8192 64} 2 > $O/pmc.out 2> $O/pmc.err
f=$(find $O/pmc -name '*counter_collection.csv' | head -1)
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    for r in csv.DictReader(open("$f")):
        k = r["Kernel_Name"][:34]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "grad_pass" not in k: continue
        print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
except Exception as e:
    print("no counters:", e)
PY
rm -rf $O/pmc
done
grep "pass" $O/pmc.out | cut -c1-120
