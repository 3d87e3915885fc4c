#!/bin/bash
# kernel stats of scratch/r5_hh_probe.py (gradient passes of cfg3 in modes f16x2 / <R3> / <HH>): $1 = tag, $2.. = probe arguments
TAG=${1:-prof_hh}; shift
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o s -- python scratch/r5_hh_probe.py "$@" > $O/out.txt 2>&1
cp $(ls $O/kt/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/kt
grep -v "^W2\|^E2\|amdgpu.ids" $O/out.txt | tail -12
head -14 $O/kernel_stats.csv | cut -c1-150
