#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_bt; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_callbacks.py tests/test_gpu_nmf.py -q -k "backtracking or unmixing or user" > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/tests.txt | head -30
