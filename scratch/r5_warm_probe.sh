#!/bin/bash
# does the 20-step line depend on how long the chain ran before the timed region?
cd $GRAFT_REPO_ROOT
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('it/s %.1f | ms %.4f | k1 %.4f | warm %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('warmup_effective')))"; }
for rep in 1 2; do
for W in 5 50 200 1000; do echo -n "steps 20 warmup $W: "; python bench.py --steps 20 --warmup $W --no-cpu 2>/dev/null | line; done
echo -n "steps 100 warmup 20: "; python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | line
echo -n "steps 400 warmup 20: "; python bench.py --steps 400 --warmup 20 --no-cpu 2>/dev/null | line
done
