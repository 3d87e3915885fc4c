#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs (one --pmc pass each) per kernel.

    python profiles/summarize_pmc.py OUT.json LABEL=path/to/*_counter_collection.csv [LABEL=...]

For every (kernel, counter): number of dispatches, number of "real" dispatches (value >= 5 % of the kernel's
maximum: chain launches that found DevStatus::halt set are ~1 us no-ops) and the mean over the real ones.
Only kernels of this library (k_*) are kept.  Values are in rocprofv3's own units (FETCH_SIZE / WRITE_SIZE: KiB).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"\b(k_[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else None


def main():
    out = {}
    for spec in sys.argv[2:]:
        label, path = spec.split("=", 1)
        vals = defaultdict(list)
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k:
                    vals[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        res = defaultdict(dict)
        for (k, c), v in sorted(vals.items()):
            mx = max(v)
            real = [x for x in v if x >= 0.05 * mx] if mx > 0 else v
            res[k][c] = {"calls": len(v), "real_calls": len(real), "avg_real": sum(real) / max(len(real), 1)}
        out[label] = res
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
