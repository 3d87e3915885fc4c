#!/bin/bash
# HBM traffic of K1 for the CURRENT build: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over the bench
# command (MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests as 64 on gfx950 -> doubled), written together with the
# hash of the kernel sources to gpurun_out/k1_traffic.json; copy that file to profiles/k1_traffic.json and bench.py prints
# the number as roofline.traffic for exactly this build.      usage: scratch/measure_traffic.sh [cfg3 [f16x2 [rows]]] ...  (rows: a row-sharded run's share, key "cfg4_rows8192/f16x2")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traffic
mkdir -p $O
cd $R
CFG=${1:-cfg3}; MODE=${2:-f16x2}; ROWS=${3:-}
RFLAG=""; [ -n "$ROWS" ] && RFLAG="--rows $ROWS"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- python bench.py --config $CFG --mode $MODE $RFLAG --steps 8 --warmup 2 --no-cpu > /dev/null 2>&1
done
python - "$CFG" "$MODE" $O "$ROWS" <<'PY'
import csv, glob, json, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
cfg, mode, O, rows = sys.argv[1:5]
key = "%s%s/%s" % (cfg, "_rows" + rows if rows else "", mode)
def avg(counter):
    path = glob.glob(os.path.join(O, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_grad" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    real = [x for x in v if x >= 0.05 * max(v)]      # launches that found the chain halted are ~1 us no-ops
    return sum(real) / len(real), len(real)
f, nf = avg("FETCH_SIZE")
w, nw = avg("WRITE_SIZE")
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "k1_traffic.json")
rec = json.load(open(out)) if os.path.exists(out) else (json.load(open(bench.TRAFFIC_FILE)) if os.path.exists(bench.TRAFFIC_FILE) else {})
rec[key] = {"source_hash": bench.kernel_source_hash(), "fetch_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024),
                            "bytes_per_launch": int(2 * f * 1024 + w * 1024), "launches_averaged": [nf, nw],
                            "when": time.strftime("%Y-%m-%d %H:%M"), "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --config %s --mode %s%s --steps 8 --warmup 2 --no-cpu; KiB -> bytes, FETCH x 2" % (cfg, mode, " --rows " + rows if rows else "")}
json.dump(rec, open(out, "w"), indent=1)
print(key, json.dumps(rec[key]))
PY
rm -rf $O
