#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3_tests
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests/pytest_gpu.txt 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/r3_tests/pytest_gpu.txt | tail -3
