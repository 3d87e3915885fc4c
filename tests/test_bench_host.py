"""bench.py's host-side pieces that need no GPU: the RCCL set-up excerpt of multi-GPU lines, the key that ties a PMC traffic
measurement to the build that prints it, the roofline arithmetic."""
import json
import os

import pytest

import bench


def test_rccl_debug_excerpt_keeps_the_lines_that_say_what_was_used(tmp_path):
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "host:12:12 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6",
        "host:12:12 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 23000 - Init START",
        "host:12:40 [0] NCCL INFO Channel 00/16 : 0 1 2 3 4 5 6 7",
        "host:12:40 [0] NCCL INFO Channel 01/16 : 0 1 2 3 4 5 6 7",
        "host:12:40 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1",
        "host:12:40 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC",
        "host:12:40 [0] NCCL INFO Connected all rings",
        "host:12:40 [0] NCCL INFO   Algorithm   |  Tree  |  Ring  |",
        "host:12:40 [0] NCCL INFO something irrelevant",
    ]))
    r = bench.rccl_debug_excerpt(str(log))
    assert r["log_lines"] == 9
    ex = r["excerpt"]
    assert any("RCCL version" in x for x in ex) and any("nranks 8" in x for x in ex) and any("via P2P" in x for x in ex)
    assert any("Algorithm" in x for x in ex) and not any("irrelevant" in x for x in ex)
    assert sum("Channel 0" in x and "/16" in x for x in ex) == 1            # repeated patterns once
    assert bench.rccl_debug_excerpt(None) is None and bench.rccl_debug_excerpt(str(tmp_path / "missing")) is None


def test_rccl_debug_is_only_switched_on_for_multi_gpu_rccl_runs(monkeypatch):
    for k in ("NCCL_DEBUG", "NCCL_DEBUG_FILE", "NCCL_DEBUG_SUBSYS", "PMX_FORCE_SHARDED", "PMX_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    assert bench.rccl_debug_on(0, 1) is None and "NCCL_DEBUG_FILE" not in os.environ
    monkeypatch.setenv("NCCL_DEBUG", "TRACE")                                # the caller's own choice stays
    assert bench.rccl_debug_on(0, 8) is None and os.environ["NCCL_DEBUG"] == "TRACE"
    monkeypatch.setenv("NCCL_DEBUG", "VERSION")                              # what this image exports
    monkeypatch.setenv("PMX_DIST_BACKEND", "gloo")
    assert bench.rccl_debug_on(0, 8) is None
    monkeypatch.delenv("PMX_DIST_BACKEND")
    path = bench.rccl_debug_on(3, 8)
    assert path and path.endswith("_3.log") and os.environ["NCCL_DEBUG"] == "INFO" and os.environ["NCCL_DEBUG_FILE"] == path
    assert "COLL" not in os.environ["NCCL_DEBUG_SUBSYS"]                     # nothing per collective
    for k in ("NCCL_DEBUG", "NCCL_DEBUG_FILE", "NCCL_DEBUG_SUBSYS"):
        monkeypatch.delenv(k, raising=False)


def test_traffic_record_matches_the_committed_sources():
    """profiles/k1_traffic.json carries the hash of the kernel sources it was measured for; bench.py prints `traffic` only
    while that is the build at hand."""
    rec = json.load(open(bench.TRAFFIC_FILE))
    h = bench.kernel_source_hash()
    assert "cfg3/f16x2" in rec
    for v in rec.values():
        assert set(("source_hash", "fetch_bytes", "write_bytes", "bytes_per_launch", "how")) <= set(v)
        assert abs(v["bytes_per_launch"] - v["fetch_bytes"] - v["write_bytes"]) <= 2        # (each rounded separately)
    e = bench.pmc_traffic("cfg3", "f16x2")
    if rec["cfg3/f16x2"]["source_hash"] != h:
        assert e is None                    # a measurement of another build is not printed
        pytest.skip("profiles/k1_traffic.json was measured for another build of the kernels: re-run scratch/measure_traffic.sh")
    assert e is not None and 1.0 < e["bytes_per_launch"] / (16384 * 16384 * 4) < 1.3


def test_roofline_entry_arithmetic():
    M = N = 16384
    K = 64
    r = bench.roofline_entry("f16x2", M, N, K, 6.0 * M * N * K, 0.3524, 25, 0.87, "k_grad_f16_v8")
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - M * N * 4 / 0.3524e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    r32 = bench.roofline_entry("f32", M, N, K, 6.0 * M * N * K, 0.8167, 25, 0.9, "k_grad_f32_pc")
    assert r32["bound"] == "mfma" and r32["unit"] == "TFLOP/s" and abs(r32["frac"] - 6.0 * M * N * K / 0.8167e-3 / 1e12 / bench.PEAK_F32_MFMA_TFLOPS) < 1e-9
