#!/bin/bash
# LDS counters of k_grad_f16_v8 per role: residual pass only / + gSt / + gA / all
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ldspmc
mkdir -p $O
cd $R
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $O/p_$tag
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p_$tag -o p -- python scratch/k1_pmc_one.py $cfg > $O/log_$tag.txt 2>&1
  python - "$O/p_$tag" "$cfg" <<'PY'
import csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = {}
for r in csv.DictReader(open(path)):
    if "k_grad_f16_v8" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("doA doS =", sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())}, "launches", len(next(iter(acc.values()))))
PY
  grep "doA=" $O/log_$tag.txt
done
