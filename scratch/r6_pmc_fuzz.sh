#!/bin/bash
# [r6] SQ counters of the K1 kernels of cfg3 (K = 64), cfg4's share (K = 128), cfg2 (K = 32) and cfg5 side by side + the long fuzz sweep
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
SQ2="SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"
run() {  # tag, bench flags
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/$1_sq1 -o f -- python bench.py $2 --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/$1_sq2 -o f -- python bench.py $2 --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
  python profiles/summarize_pmc.py $O/pmc_$1.json sq1=$(ls $O/$1_sq1/*counter_collection.csv) sq2=$(ls $O/$1_sq2/*counter_collection.csv) 2>&1 | tail -2
  rm -rf $O/$1_sq1 $O/$1_sq2
}
run cfg3 "--config cfg3"
run cfg4rows8192 "--config cfg4 --rows 8192"
run cfg2r "--config cfg2 --mode f16x2r"
run cfg5 "--config cfg5"
python - $O <<'PY'
import json, sys, os
O = sys.argv[1]
for tag in ("cfg3", "cfg4rows8192", "cfg2r", "cfg5"):
    try:
        d = json.load(open(os.path.join(O, "pmc_%s.json" % tag)))
    except Exception as e:
        print(tag, "ERR", e); continue
    allk = {}
    for lab in d:
        for k, v in d[lab].items():
            if "k_grad" in k:
                allk.setdefault(k, {}).update({c: x["avg_real"] for c, x in v.items()})
    for k, v in allk.items():
        wc = v.get("SQ_WAVE_CYCLES", 1)
        print(tag, k[:60])
        for c in sorted(v):
            print("     %-28s %14.0f  (%.3f of wave cycles)" % (c, v[c], v[c] / wc))
PY
python scratch/r6_fuzz.py > $O/fuzz.txt 2>&1; tail -12 $O/fuzz.txt
