"""The product path needs no torch: a fresh interpreter that never imports it runs a context, and the multi-GPU collectives of
the C ABI (pmx_comm_*: RCCL looked up at run time -- here the ROCm installation's own librccl.so.1, nothing loaded it before)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_context_and_collectives_without_torch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "no_torch_process.py")], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "comm without torch ok" in r.stdout
