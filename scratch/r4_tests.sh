#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_tests; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/tests.txt | head -40
cp gpurun_out/parity_fractions.json gpurun_out/parity_long.json $O/ 2>/dev/null
