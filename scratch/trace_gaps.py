# scratch: per-iteration timeline from a rocprofv3 kernel trace (durations and gaps between consecutive kernels)
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    m = re.search(r"\b(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
    name = m.group(1) if m else ("nccl" if "ccl" in r["Kernel_Name"].lower() else None)
    if name:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
# find K1 launches that really ran (> 100 us) and print the timeline of the last few iterations
k1 = [i for i, e in enumerate(ev) if e[2].startswith("k_grad") and e[1] - e[0] > (int(sys.argv[2]) if len(sys.argv) > 2 else 100000)]
for a, b in zip(k1[-4:-1], k1[-3:]):
    print("--- iteration: %.1f us from K1 start to next K1 start" % ((ev[b][0] - ev[a][0]) / 1e3))
    prev_end = None
    for e in ev[a:b]:
        gap = (e[0] - prev_end) / 1e3 if prev_end else 0.0
        print("   %-18s dur %8.1f us   gap before %6.1f us" % (e[2], (e[1] - e[0]) / 1e3, gap))
        prev_end = e[1]
    print("   (gap to next K1 %.1f us)" % ((ev[b][0] - prev_end) / 1e3))
