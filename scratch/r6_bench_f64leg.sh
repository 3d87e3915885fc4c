#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6f64
SECONDS=0; python bench.py --steps 20 --warmup 5 > gpurun_out/r6f64/bench_20_5.json 2> gpurun_out/r6f64/bench_20_5.err
echo "bench.py wall: $SECONDS s"
grep "^{" gpurun_out/r6f64/bench_20_5.json | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d.get("fp64_inputs"), indent=1)[:1500])
print({k: v for k, v in d.get("measured", {}).items() if "fp64" in k})
'
