cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/small
mkdir -p $O
cd $R
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o s -- python scratch/small_trace.py > $O/log.txt 2>&1
python scratch/trace_gaps.py $(ls $O/t/*kernel_trace.csv) 20000 > $O/timeline.txt 2>/dev/null
head -40 $O/timeline.txt
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/small/t/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# print a window of 14 consecutive kernels from the middle of the pgm run and of the adaprox run
names = [r["Kernel_Name"][:60] for r in rows]
def show(i0):
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:i0 + 14]:
        print("%8.1f us  dur %6.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60]))
idx = [i for i, n in enumerate(names) if "k_grad_small" in n]
print(len(idx), "k_grad_small launches")
show(idx[30]); print(); show(idx[100])
PY
find $O/t -name "*kernel_trace.csv" -delete
