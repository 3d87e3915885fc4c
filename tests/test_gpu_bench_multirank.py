"""The driver's multi-GPU command line, end to end: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.
One GPU per box here, so the two ranks share device 0 and talk over gloo (test-only overrides PMX_DIST_BACKEND /
PMX_BENCH_DEVICE; RCCL needs a device per rank): what is checked is the launch contract -- rendezvous on 127.0.0.1, every
back-end's sharded leg, ONE JSON line from rank 0 with the whole-job rate -- not the numbers."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("extra", [["--rows", "2048"], ["--config", "cfg4", "--rows", "2048"], ["--config", "cfg2", "--rows", "1024"],
                                   ["--config", "cfg5", "--rows", "2048"]])
def test_bench_with_two_ranks(extra):
    env = dict(os.environ, PMX_DIST_BACKEND="gloo", PMX_BENCH_DEVICE="0", PMX_TAIL_FUSED="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 alone prints
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1.0        # whole-job iterations per second = 1 / (max-over-ranks time per step)
    assert "sharded over 2 GPUs" in d["config"]["parallelism"]
