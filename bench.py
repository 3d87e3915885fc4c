#!/usr/bin/env python3
"""bench.py -- NMF iterations/sec on MI355X for BASELINE.json's headline configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg5] [--no-cpu]

A "step" is ONE solver iteration of proxmin_amd's nmf() hot path on synthetic data already resident
in HBM: the fused residual-gradient pass over Y (6MNK flop; 8MNK and two passes for bsdmm) plus the
factor update, proximal operators and stopping-test reductions.  Default workload = BASELINE.json
configs[2] ("cfg3"): Y 16384 x 16384, K = 64, adaprox/AMSGrad, prox_plus on A, prox_unity_plus on
the columns of S.  With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the rows
of Y and A are sharded over the ranks and gS is all-reduced once per iteration (strong scaling).

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (K1, k_grad_f32) from HIP
events recorded on its launch stream inside the timed region; `cpu_baseline` is the NumPy oracle
(a validated restatement of the reference) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _physical_cores():
    try:
        import psutil
        return int(psutil.cpu_count(logical=False) or 0) or None
    except Exception:
        return None


# the CPU leg (BLAS threads = all physical cores): asked for BEFORE NumPy loads its BLAS, which reads the
# variable once.  A wheel whose OpenBLAS was built with a lower thread limit (NumPy's: 64) clamps it; cpu_baseline() reports
# what the library really runs with next to what was asked.
if _physical_cores():
    os.environ.setdefault("OPENBLAS_NUM_THREADS", str(_physical_cores()))

import numpy as np  # noqa: E402

CONFIGS = {
    # name: (M, N, K, backend, unity_S, description)
    "cfg2": (4096, 4096, 32, "pgm", False, "Y 4096x4096, K=32, prox_plus, PGM, fp32"),
    "cfg3": (16384, 16384, 64, "adaprox", True, "Y 16384x16384, K=64, adaprox/AMSGrad, prox_plus(A) + prox_unity_plus(S columns)"),
    "cfg4": (65536, 16384, 128, "adaprox", False, "Y 65536x16384, K=128, adaprox/AMSGrad, prox_plus (BASELINE's 8-GPU case)"),
    "cfg5": (16384, 16384, 64, "bsdmm", False, "Y 16384x16384, K=64, bSDMM, proxs_g=[prox_plus, prox_soft(1e-3)] per factor"),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.3 TB/s is what a float4 copy achieves)
MODE_DTYPE = {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA, fp32 accumulate, fp32-class accuracy)",
              "f16x2": "f16x2 (two-term fp16 split MFMA, fp32 accumulate, fp32-class accuracy)",
              "f16x2r": "f16x2r (fp16 split MFMA, fp32 accumulate: the residual from the high x high product + an exact K x K correction, two-term gradients; exact fp32's error class)"}
MODE_DESC = {"f32": "f32 (exact fp32 MFMA, Y fp32 in HBM)",
             "bf16x3": "bf16x3 (operands split into bf16 terms: 6 MFMA passes for A@S, 3 for each gradient; fp32 accumulate; Y fp32 in HBM)",
             "f16x2": "f16x2 (operands scaled by powers of two and split into two fp16 terms: 3 MFMA passes for each of A@S and the "
                      "two gradients; fp32 accumulate; Y fp32 in HBM)",
             "f16x2r": "f16x2r (operands scaled by powers of two and split into fp16 terms; A@S from the HIGH terms alone -- one MFMA pass -- and what that leaves out, "
                       "A s_r + a_r s0, restored exactly through K x K matrices (k_gfix.hip); three passes for each gradient: 1 + 3 + 3 MFMA passes per 3 contractions; "
                       "fp32 accumulate; Y fp32 in HBM; gradients in exact fp32's error class)"}
# issued MFMA flops per algorithmic flop (products per 3 contractions / 3): informational -- issued flops earn no roofline credit
MFMA_PASSES = {"bf16x3": 4.0, "f16x2": 3.0, "f16x2r": 11.0 / 3.0, "f16x2g": 7.0 / 3.0}


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "k1_traffic.json")


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the kernel sources libpmx.so is built from: the key that ties a PMC measurement
    (profiles/k1_traffic.json, written by scratch/measure_traffic.sh from rocprofv3 --pmc passes over this very command) to
    the build that prints it."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "proxmin_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and f != "bench_floor.hip":     # (the measurement skeleton is not a kernel of the product)
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(config, mode):
    """HBM bytes per K1 launch (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE) measured for THIS build, else None:
    a number from another build would go stale silently."""
    try:
        rec = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None
    e = rec.get("%s/%s" % (config, mode))
    if not e or e.get("source_hash") != kernel_source_hash():
        return None
    return e


def effective_mode(dev):
    """Arithmetic the context's K1 really runs in: a split-precision mode falls back to exact fp32 where it has no kernel."""
    return "f32" if dev.k1_info()["kernel"] in ("k_grad_f32", "k_grad_f32_pc") else dev.mode


def roofline_entry(mode, M, N, K, flop_per_launch, k1_avg_ms, k1_n, share, kernel=None, both=True):
    """Dominant kernel = K1 (fused residual-gradient), priced on ALGORITHMIC work: 6 M N K flop and one pass over Y,
    M N 4 bytes, per launch.  Mode f32: the exact-fp32 MFMA peak bounds it.  Split modes: the algorithmic intensity 6 K / 4 flop per
    byte (48 / 96 / 192 at K = 32 / 64 / 128) times 8 TB/s is below the dense bf16 / fp16 MFMA peak at every K <= 128, so the pass
    over Y is the roof: frac = (M N 4 bytes / launch time) / 8 TB/s.  Issued MFMA flops (several fp16 products per fp32-class
    multiply-add) are reported for information only."""
    t = k1_avg_ms * 1e-3
    tflops = flop_per_launch / t / 1e12
    gbs = (M * N * 4) / t / 1e9
    kp = 32 if K <= 32 else 64 if K <= 64 else 128
    if mode == "f32":
        return {"kernel": "k_grad_f32_pc<%d>" % K if kernel == "k_grad_f32_pc" else "k_grad_f32<%d>" % kp, "bound": "mfma", "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": None, "avg_launch_ms": k1_avg_ms,
                "launches": k1_n, "hbm_gbs_algorithmic": gbs, "k1_share_of_step": share}
    names = {"k_grad_f16_v8": ("k_grad_f16_v8", "f16x2"), "k_grad_f16_v8_r3": ("k_grad_f16_v8<R3>", "f16x2r"), "k_grad_f16_v8_hh": ("k_grad_f16_v8<HH> + k_gfix", "f16x2g"),
             "k_grad_f16_k32": ("k_grad_f16_k32", "f16x2"), "k_grad_f16_k32_r3": ("k_grad_f16_k32<R3>", "f16x2r"),
             "k_grad_f16_k128": ("k_grad_f16_k128", "f16x2"), "k_grad_f16_k128_hh": ("k_grad_f16_k128<HH> + k_gfix", "f16x2g")}
    name, arith = names.get(kernel, ("k_grad_bf16_v7" if (K == 64 and M % 128 == 0 and N % 256 == 0) else "k_grad_bf16<%d>" % kp, "bf16x3"))
    if both and kernel in ("k_grad_f16_v8_hh", "k_grad_f16_k128_hh") and os.environ.get("PMX_K1_ROLE_SPLIT", "1") != "0":
        name = name.replace("<HH>", "<HH, RS>")      # passes that want both gradients: the consumers' roles split by contraction (k_grad_f16_v8.hip: RS)
    passes = MFMA_PASSES[arith]
    return {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": gbs / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": k1_avg_ms, "launches": k1_n,
            "algorithmic_tflops": tflops, "algorithmic_frac_of_bf16_mfma_peak": tflops / PEAK_BF16_MFMA_TFLOPS,
            "hbm_roof_tflops": 6.0 * K / 4.0 * PEAK_HBM_GBS / 1e3,      # what one pass over fp32 Y allows: 6K/4 flop/B x 8 TB/s
            "mfma_issued_tflops": passes * tflops, "mfma_products_per_3_contractions": round(3 * passes), "k1_share_of_step": share}


def measured_on_this_box(local):
    """[r6] SURVEY 8(d): "the measured (not nominal) peaks ... on the box", next to the nominal ones the roofline is priced against, and the ENERGY
    FLOOR of K1's own job: the skeleton of k_grad_f16_v8<HH, RS> (its grid, its pass over a 16384 x 16384 fp32 Y, its 28 fp16 MFMAs per SIMD and
    128 x 32 block, a barrier per block -- and nothing else) timed on this package under its power cap (proxmin_amd/csrc/bench_floor.hip, built by
    __graft_entry__.build() into libpmx_floor.so).  All on random data: the package clocks to its power budget and zeros run up to 19 % faster
    (MI355X_MICROARCH.md, DVFS)."""
    import ctypes as C
    path = os.path.join(ROOT, "proxmin_amd", "libpmx_floor.so")
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        return {"error": "libpmx_floor.so not loadable: %r" % (exc,)}
    lib.pmxf_last_error.restype = C.c_char_p
    lib.pmxf_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.pmxf_copy.argtypes = [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double)]
    lib.pmxf_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.pmxf_mfma_f64.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = {}

    def call(fn, *a):
        v = C.c_double()
        rc = fn(*a, C.byref(v))
        if rc != 0:
            raise RuntimeError(lib.pmxf_last_error().decode("utf-8", "replace"))
        return v.value
    try:
        M = N = 16384
        gb = M * N * 4 / 1e9
        out["hbm_copy_gbs"] = call(lib.pmxf_copy, local, 1 << 30, 10)                    # float4 copy kernel, bytes read + written per second
        ms = {}
        # variant = fetch (0: 4 B / lane, 1: 8 B / lane pairs = K1's fetch, 2: 16 B / lane, 3: none) + 10 ops (random fragments from LDS) + 100 epilogue + 1000 MFMAs
        for name, var in (("stream_8B", 1), ("stream_16B", 2), ("mfma28_const", 1003), ("mfma28_random_lds", 1013),
                          ("stream_8B+mfma28_const", 1001), ("stream_16B+mfma28_const", 1002),
                          ("stream_8B+mfma28_random_lds", 1011), ("stream_16B+mfma28_random_lds", 1012),
                          ("stream_8B+mfma28_random_lds+epilogue", 1111), ("stream_16B+mfma28_random_lds+epilogue", 1112),
                          ("stream_8B+mfma36_const (round 3's arithmetic)", 2001)):
            ms[name] = call(lib.pmxf_stream, local, var, M, N, 0, 20)
        out["hbm_read_gbs"] = gb / (ms["stream_16B"] * 1e-3)                              # K1's grid reading Y once, 16 B per lane, nothing else
        out["hbm_read_gbs_8B_per_lane"] = gb / (ms["stream_8B"] * 1e-3)
        out["fp16_mfma_tflops_random"] = call(lib.pmxf_mfma, local, 1, 10)                # dense v_mfma_f32_32x32x16_f16 from registers, random operands
        out["fp16_mfma_tflops_zeros"] = call(lib.pmxf_mfma, local, 0, 10)
        out["fp64_mfma_tflops_random"] = call(lib.pmxf_mfma_f64, local, 1, 2, 5)          # dense v_mfma_f64_16x16x4_f64, two waves per SIMD (the fp64 path's roof: fp64_inputs)
        out["fp64_mfma_tflops_random_one_wave_per_simd"] = call(lib.pmxf_mfma_f64, local, 1, 1, 5)
        out["skeleton_ms"] = ms
        # the floor K1 is priced against: K1's own fetch shape and operands that toggle like real data, no epilogue (the epilogue variant is what a
        # kernel that ALSO forms the residual cannot avoid; both are printed)
        out["energy_floor_ms"] = ms["stream_8B+mfma28_random_lds"]
        out["energy_floor_ms_16B_fetch"] = ms["stream_16B+mfma28_random_lds"]
        out["energy_floor_with_epilogue_ms"] = ms["stream_8B+mfma28_random_lds+epilogue"]
        # [r6] the same skeleton for the two other K1 shapes the line quotes: cfg4's 8192-row share at K = 128 (8 + 48 MFMAs per producer + consumer wave and block:
        # the HBM roof is the wrong ruler at 192 flop per byte) and a one-gradient pass at K = 64 (bsdmm: 4 + 12)
        try:
            out["energy_floor_k128_share8192_ms"] = call(lib.pmxf_stream, local, 3011, 8192, 16384, 0, 20)
            out["mfma56_random_lds_share8192_ms"] = call(lib.pmxf_stream, local, 3013, 8192, 16384, 0, 20)
            out["energy_floor_one_gradient_pass_ms"] = call(lib.pmxf_stream, local, 4011, M, N, 0, 20)
        except Exception as exc:                  # (an older libpmx_floor.so without these variants)
            out["energy_floor_other_shapes_error"] = repr(exc)
        out["note"] = ("measured on this GPU just before the headline, random data, 2 x 20 launches each after a warm pass; skeleton = K1's grid / region map / barrier per "
                       "128 x 32 block / 4 + 24 MFMAs per producer + consumer wave and block, no gradient computed (proxmin_amd/csrc/bench_floor.hip)")
    except Exception as exc:                      # a side measurement must never take the headline down
        out["error"] = repr(exc)
    return out


def make_problem_device(M, N, K, unity, seed, device):
    """Seeded synthetic Y = A_true S_true + 0.01 noise generated ON THE GPU (fp32; torch.Generator(seed): SURVEY 8(d)'s recipe and
    statistics, not the numbers np.random.default_rng(1234) would draw for A_true / S_true / the noise -- a 1 GiB Y drawn on the
    host would add ~10 s and a PCIe copy to every run), A0 / S0 from np.random.default_rng(seed) on the host.  GENERATOR says so in
    the JSON line."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    At = torch.rand((M, K), generator=g, device=device, dtype=torch.float32)
    St = torch.rand((K, N), generator=g, device=device, dtype=torch.float32)
    if unity:
        St /= St.sum(0, keepdim=True)
    Y = At @ St
    Y += 0.01 * torch.randn((M, N), generator=g, device=device, dtype=torch.float32)
    rng = np.random.default_rng(seed)
    A0 = rng.random((M, K), dtype=np.float32)
    S0 = rng.random((K, N), dtype=np.float32)
    if unity:
        S0 /= S0.sum(0, keepdims=True)
    return Y, A0, S0


GENERATOR = ("Y = A_true @ S_true + 0.01 N(0,1), A_true / S_true uniform [0,1) (S_true columns normalised where the config has prox_unity), drawn on the GPU "
             "by torch.Generator(seed 1234); A0, S0 from np.random.default_rng(1234) on the host (SURVEY 8(d)'s recipe; the tests draw everything with NumPy)")


def begin_solver(dev, backend, unity):
    from functools import partial
    from proxmin_amd import operators as ops
    pA = ops.device_proxseq(ops.prox_plus, 0)
    pS = ops.device_proxseq(partial(ops.prox_unity_plus, axis=0) if unity else ops.prox_plus, 1)
    if backend == "pgm":
        dev.pgm_begin([pA, pS], accelerated=False, e_rel=(1e-12, 1e-12))
        return lambda n: dev.pgm_run(n)
    if backend == "adaprox":
        # nmf() defaults: e_rel = 1e-3 feeds the proximal sub-iteration test; check_convergence off so
        # that the outer loop runs a fixed number of iterations
        dev.adaprox_begin([pA, pS], scheme="amsgrad", check_convergence=False, prox_max_iter=1000, e_rel=(1e-3, 1e-3))
        return lambda n: dev.adaprox_run(np.full(n, 0.9), 0.9)
    pg = [ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_soft, thresh=1e-3), 0)]
    dev.bsdmm_begin([pA, pS], [pg, pg], e_rel=(1e-12, 1e-12), e_abs=(0.0, 0.0))
    return lambda n: dev.bsdmm_run(n)


def cpu_baseline(Y, A0, S0, backend, unity, n_iter=6, transient=0):
    """Oracle (NumPy port of the reference, fp32 like the device data) on this host's cores: the SAME workload at its full
    size -- the bench's own Y (copied back from the GPU) and initial factors.  `transient` iterations run untimed first (the
    adaprox start-up: 250 proximal passes on S in iteration 0, tens in the next few), then `n_iter` iterations are timed from
    callback time stamps: the same steady state the GPU's timed region is in.  Returns the record
    and the per-iteration proximal pass counts (A, S) of the timed iterations."""
    from oracle import nmf_oracle as orc
    M, N = Y.shape
    K = A0.shape[1]
    A, S = A0.copy(), S0.copy()
    stamps = []
    total = transient + n_iter

    def cb(*X, it=None):
        stamps.append(time.perf_counter())

    sub_note, sub_timed = "", None
    if backend == "pgm":
        orc.pgm_nmf(Y, A, S, max_iter=total, e_rel=1e-12, callback=cb)
    elif backend == "adaprox":
        # sub-iteration counts of the timed window alone: the run is split at the window's start (warm moments carried over)
        ret = orc.adaprox_nmf(Y, A, S, ("plus",), ("unity_plus", 0) if unity else ("plus",), scheme="amsgrad",
                              max_iter=total, e_rel=1e-3, check_convergence=False, callback=cb,
                              sub_trace=True)
        per = ret[6]
        sub_timed = [float(np.mean([p[j] for p in per[transient:]])) for j in range(2)]
        sub_note = "; proximal passes per iteration inside the timed window: A %.2f, S %.2f (iteration 0: A %d, S %d)" % (
            sub_timed[0], sub_timed[1], per[0][0], per[0][1])
    else:
        orc.bsdmm_nmf(Y, A, S, proxs_g=[[("plus",), ("soft", 1e-3, "relative")]] * 2, max_iter=total, e_rel=1e-12, callback=cb)
    stamps.append(time.perf_counter())
    per_t = np.diff(stamps)[max(transient, 1):]            # the timed window (never iteration 0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    blas = "?"
    host_cores = cores
    try:
        from threadpoolctl import threadpool_info
        infos = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        blas = ", ".join("%s %s, %d threads" % (i.get("internal_api"), i.get("version"), i.get("num_threads")) for i in infos) or "?"
        if infos:
            cores = max(int(i.get("num_threads") or 1) for i in infos)     # the threads the contractions really use (the rest of NumPy is one thread)
    except Exception:
        pass
    phys = None
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        pass
    rec = {"value": 1.0 / float(np.mean(per_t)), "unit": "it/s", "cores": cores, "kind": "port",
           "sample": "oracle (NumPy fp32; BLAS: %s; OPENBLAS_NUM_THREADS=%s asked [= physical cores; the library's own build limit clamps it: `cores` is what it runs with]; host: %d hardware threads, %s physical cores) on the full %d x %d x %d workload "
                     "(the bench's own Y and initial factors): %d untimed iterations (start-up transient), then %d timed, mean %.3f s (min %.3f, max %.3f), no scaling%s"
                     % (blas, os.environ.get("OPENBLAS_NUM_THREADS", "unset"), host_cores, phys, M, N, K, transient, len(per_t),
                        float(np.mean(per_t)), float(per_t.min()), float(per_t.max()), sub_note)}
    return rec, sub_timed


RCCL_LOG = "/tmp/pmx_bench_rccl_%d.log"


def rccl_debug_on(rank, world):
    """Multi-GPU runs: have RCCL write what it decided at communicator set-up (topology, rings / trees, transports, the
    algorithm / protocol tuning table) to a per-rank file -- INIT, GRAPH and TUNING only, nothing per collective -- so that the
    JSON line can say over WHAT the whole-job number was measured.  An NCCL_DEBUG level the caller chose (INFO, TRACE) is left alone;
    VERSION / WARN (this image exports VERSION) are raised to INFO, into the file."""
    if (world <= 1 and not os.environ.get("PMX_FORCE_SHARDED")) or os.environ.get("NCCL_DEBUG", "VERSION").upper() not in ("VERSION", "WARN") or os.environ.get("PMX_DIST_BACKEND", "nccl") != "nccl":
        return None
    path = RCCL_LOG % rank
    try:
        if os.path.exists(path):
            os.remove(path)
    except OSError:
        return None
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING"
    os.environ["NCCL_DEBUG_FILE"] = path
    return path


def rccl_debug_excerpt(path, limit=16):
    """A few lines of rank 0's RCCL set-up log: version, ranks, channels, transports, algorithm / protocol table."""
    import re
    if not path:
        return None
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError:
        return None
    pat = re.compile(r"(RCCL version|NCCL version|nranks|Channel \d+/\d+ *:|Ring \d+ *:|Trees? |via P2P|via SHM|via NET|[Cc]onnected all|Algorithm|Protocol|AllReduce|ReduceScatter|AllGather|xGMI|XGMI|nChannels)")
    keep, seen = [], set()
    for ln in lines:
        if pat.search(ln):
            body = ln.split("NCCL INFO", 1)[-1].strip()[:200]
            key = re.sub(r"\d+", "#", body)[:60]
            if key in seen:
                continue
            seen.add(key)
            keep.append(body)
            if len(keep) >= limit:
                break
    return {"log_lines": len(lines), "excerpt": keep}


def emit(out):
    """The ONE JSON line, as the LAST line of stdout: libraries that write through C stdio (RCCL's version banner) have
    their buffer flushed first, and the process leaves without running exit-time destructors that could print more."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def other_configs(Y3, local):
    """Short runs of the other BASELINE configurations in the same process as the headline (one line each: it/s, K1 time from
    HIP events, roofline fraction of K1): cfg2 (4096^2, K=32, PGM, exact fp32), cfg5 (cfg3's Y, bSDMM, f16x2), one rank's
    8192-row share of cfg4 (K=128, adaprox, f16x2; the first 8192 rows of cfg3's Y serve as its data) on the single-GPU code
    path.  Bounded to a few seconds."""
    import torch
    from proxmin_amd.engine import DeviceNMF
    device = Y3.device
    res = {}
    specs = [("cfg2", "f32", 4096, 4096, 32, "pgm", False, 200, 40, None),
             ("cfg2_f16x2", "f16x2", 4096, 4096, 32, "pgm", False, 200, 40, None),      # the same problem in the headline's arithmetic (k_grad_f16_k32)
             ("cfg2_f16x2r", "f16x2r", 4096, 4096, 32, "pgm", False, 200, 40, None),    # ... and with the residual in exact fp32's class (k_grad_f16_k32<R3>)
             ("cfg5", "f16x2r", 16384, 16384, 64, "bsdmm", False, 30, 10, Y3),
             ("cfg4_share8192", "f16x2r", 8192, 16384, 128, "adaprox", False, 40, 20, Y3)]
    for name, mode, M, N, K, backend, unity, steps, warm, Yuse in specs:
        try:
            if Yuse is None:
                Yc, A0, S0 = make_problem_device(M, N, K, unity, 1234, device)
            else:
                Yc = Yuse
                rng = np.random.default_rng(4321)
                A0 = rng.random((M, K), dtype=np.float32)
                S0 = rng.random((K, N), dtype=np.float32) * np.float32(16.0 / K)    # A0 @ S0 on the scale of cfg3's Y
            dev = DeviceNMF(M, N, K, device=local, mode=mode)
            dev.set_Y_device(Yc.data_ptr(), ld=N, copy=False, keepalive=Yc)
            dev.set_factors(A0, S0)
            run = begin_solver(dev, backend, unity)
            run(warm)
            dev.set_timing(True, every=4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            k1_ms, k1_n = dev.get_timing()
            dev.set_timing(False)
            assert r.iterations == steps
            nk1 = 2 if backend == "bsdmm" else 1
            flop_launch = (8.0 if backend == "bsdmm" else 6.0) * M * N * K / nk1
            k1_avg = k1_ms / max(k1_n, 1)
            info = dev.k1_info()
            roof = roofline_entry(effective_mode(dev), M, N, K, flop_launch, k1_avg, k1_n, k1_avg * nk1 * steps / (1e3 * dt), info["kernel"], backend != "bsdmm")
            tr = pmc_traffic({"cfg4_share8192": "cfg4_rows8192", "cfg2_f16x2": "cfg2", "cfg2_f16x2r": "cfg2"}.get(name, name), effective_mode(dev))
            res[name] = {"value": steps / dt, "unit": "it/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warm,
                         "mode": mode, "backend": backend, "shape": [M, N, K], "k1_kernel": roof["kernel"] + ("<chain %d>" % info["chain"] if info["chain"] else ""),
                         "k1_ms": k1_avg, "tail_ms": 1e3 * dt / steps - nk1 * k1_avg,      # the step minus its K1 launches: update kernels, step rule, gaps
                         "roofline_bound": roof["bound"], "roofline_frac": roof["frac"], "k1_share_of_step": roof["k1_share_of_step"],
                         # HBM bytes per K1 launch from rocprofv3 PMC passes over THIS build (profiles/k1_traffic.json, hash-checked), else null
                         "traffic": tr["bytes_per_launch"] if tr else None, "algorithmic_bytes": M * N * 4,
                         "traffic_over_algorithmic": (tr["bytes_per_launch"] / float(M * N * 4)) if tr else None}
            dev.close()
            del dev
        except Exception as exc:                                   # a side measurement must never take the headline down
            res[name] = {"error": repr(exc)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="the headline alone: no CPU baseline and none of the side legs (peaks, other configurations, nmf() leg)")
    ap.add_argument("--skip-cpu-baseline", action="store_true", help="everything but the CPU leg (tuning: A/B runs that want other_configs)")
    ap.add_argument("--rows", type=int, default=0, help="override M (debug)")
    ap.add_argument("--mode", default=None, choices=["f32", "bf16x3", "f16x2", "f16x2r"],
                    help="contraction arithmetic: f16x2r = fp16 split MFMA in exact fp32's error class (default for cfg3 / cfg4 / cfg5: the headline), "
                         "f16x2 = two-term fp16 split MFMA (faster to issue at K = 32 only; 2-4 x fp32's out-of-tolerance entries), "
                         "bf16x3 = three-term bf16 split MFMA, f32 = exact fp32 MFMA (default for cfg2, which BASELINE quotes in fp32)")
    args = ap.parse_args()
    if args.mode is None:
        # [r6] the LIBRARY'S default arithmetic (proxmin_amd.engine.LIBRARY_DEFAULT_MODE = f16x2r): the headline needs no --mode; cfg2 is the one
        # configuration BASELINE quotes in fp32, so its own line runs exact fp32 (its f16x2r number is in other_configs)
        from proxmin_amd.engine import LIBRARY_DEFAULT_MODE
        args.mode = "f32" if args.config == "cfg2" else LIBRARY_DEFAULT_MODE

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rccl_log = rccl_debug_on(rank, world)     # before torch (and with it RCCL) is loaded: the library reads its environment once
    import torch
    import __graft_entry__ as g
    if "PMX_BENCH_DEVICE" in os.environ:      # test-only: several ranks on one GPU (see proxmin_amd/distributed.py)
        local = int(os.environ["PMX_BENCH_DEVICE"])
    torch.cuda.set_device(local)              # before any collective: RCCL binds a rank to the current device
    device = torch.device("cuda", local)
    if world > 1:
        # rank 0 (re)builds the library if it is stale; nobody loads it before that is done
        import torch.distributed as dist
        dist.init_process_group(backend=os.environ.get("PMX_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
        if rank == 0:
            g.build()
        dist.barrier()
    else:
        g.build()
    M, N, K, backend, unity, desc = CONFIGS[args.config]
    if args.rows:
        M = args.rows

    if world > 1 or os.environ.get("PMX_FORCE_SHARDED"):
        import bench_multi
        args.mode_dtype, args.mode_desc = MODE_DTYPE, MODE_DESC
        out = bench_multi.bench_sharded(args, M, N, K, backend, unity, desc, rank, world, local)
        if rank == 0:
            if rccl_log:
                out["rccl"] = rccl_debug_excerpt(rccl_log)
            emit(out)
        return

    from proxmin_amd.engine import DeviceNMF
    import proxmin_amd
    Y, A0, S0 = make_problem_device(M, N, K, unity, 1234, device)
    torch.cuda.synchronize()

    # [r6] Everything that is NOT the headline runs FIRST: the box's measured peaks and K1's energy floor, the same workload in the
    # other arithmetic modes, the other BASELINE configurations, one nmf() call through the public entry point.  The timed region of the headline
    # (20 steps = 8 ms with the driver's flags: inside one DVFS time constant) then starts on a package that has been busy for seconds.
    side = {}
    full = args.config == "cfg3" and not args.rows and not args.no_cpu
    if full:
        side["measured"] = measured_on_this_box(local)
        for key, mode, warm_s, steps_s, note in (("value_f32_mode", "f32", 25, 20, "same workload in exact fp32 (mode f32: %s)"),
                                                 ("value_f16x2_mode", "f16x2", 25, 40, "same workload in mode f16x2 (%s: two fp16 terms per operand in all three contractions; noisier than exact fp32: "
                                                                                       "2-4 x its out-of-tolerance entries against the fp64 oracle, tests/test_gpu_parity_long.py)")):
            try:
                devs = DeviceNMF(M, N, K, device=local, mode=mode)
                devs.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
                devs.set_factors(A0, S0)
                runs = begin_solver(devs, backend, unity)
                runs(warm_s)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                runs(steps_s)
                torch.cuda.synchronize()
                side[key] = {"value": steps_s / (time.perf_counter() - t0), "unit": "it/s", "steps": steps_s, "warmup": warm_s, "note": note % devs.k1_info()["kernel"]}
                devs.close()
            except Exception as exc:
                side[key] = {"error": repr(exc)}
        side["other_configs"] = other_configs(Y, local)
        for cfg_name, key in (("cfg4_share8192", "energy_floor_k128_share8192_ms"), ("cfg5", "energy_floor_one_gradient_pass_ms")):
            oc, fl = side["other_configs"].get(cfg_name, {}), side["measured"].get(key)
            if fl and oc.get("k1_ms"):           # K1 against the floor of ITS job (stream + its own MFMA count on random operands, under this package's cap)
                oc["energy_floor_ms"] = fl
                oc["frac_of_energy_floor"] = fl / oc["k1_ms"]
        side["nmf_call"] = nmf_call_leg(Y, A0, S0, unity)

    assert proxmin_amd.get_default_mode() == proxmin_amd.LIBRARY_DEFAULT_MODE or os.environ.get("PMX_MODE"), "bench.py must not change the library's default mode"
    dev = DeviceNMF(M, N, K, device=local, mode=args.mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = begin_solver(dev, backend, unity)

    # Untimed warm-up: the W iterations asked for, then -- adaprox only -- further iterations in groups of 5 until the
    # proximal sub-iteration loops have left their start-up transient (cold start: 250 passes on S in iteration 0, tens in
    # the next few, 2-5 in steady state), so that the timed region measures the same steady state whatever --warmup was.
    res_w = run(args.warmup) if args.warmup > 0 else None
    warm_total = args.warmup
    sub_seen = [int(res_w.sub_iterations[0]), int(res_w.sub_iterations[1])] if res_w is not None else [0, 0]
    sub_window = None                      # proximal passes per iteration (A, S) over the LAST five warm-up iterations
    if backend == "adaprox":
        while warm_total < 60:
            r = run(5)
            warm_total += 5
            sub_window = [(int(r.sub_iterations[j]) - sub_seen[j]) / 5.0 for j in range(2)]
            per_it = sub_window[1]
            sub_seen = [int(r.sub_iterations[0]), int(r.sub_iterations[1])]
            if per_it <= 6.0 and warm_total >= 20:
                break
    dev.set_timing(True, every=4)   # HIP events around every 4th K1 launch of the timed region
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(args.steps)                  # returns after the stream is idle
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    k1_ms, k1_n = dev.get_timing()
    dev.set_timing(False)
    assert res.iterations == args.steps, "chain ended early (%d of %d iterations)" % (res.iterations, args.steps)
    sub_timed = [(int(res.sub_iterations[j]) - sub_seen[j]) / float(args.steps) for j in range(2)]

    dt = t1 - t0
    flop_per_it = (8.0 if backend == "bsdmm" else 6.0) * M * N * K
    its = args.steps / dt
    k1_avg_ms = k1_ms / max(k1_n, 1)
    flop_per_launch = flop_per_it / (2 if backend == "bsdmm" else 1)   # bsdmm: two K1 launches of 4MNK each
    out = {
        "metric": "NMF iterations/sec at Y=%dx%d, K=%d" % (M, N, K),
        "value": its, "unit": "it/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "warmup_effective": warm_total,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": MODE_DTYPE[effective_mode(dev)], "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.config, desc), "M": M, "N": N, "K": K, "backend": backend,
                   "mode": MODE_DESC[effective_mode(dev)], "mode_is_library_default": dev.mode == proxmin_amd.LIBRARY_DEFAULT_MODE,
                   "parallelism": "1 GPU", "generator": GENERATOR},
        "gflops": flop_per_it * its / 1e9,
        "sub_iterations_per_step": sub_timed,     # proximal passes per iteration (A, S) inside the timed region only
        "roofline": roofline_entry(effective_mode(dev), M, N, K, flop_per_launch, k1_avg_ms, k1_n,
                                   k1_avg_ms * (2 if backend == "bsdmm" else 1) * args.steps / (1e3 * dt), dev.k1_info()["kernel"], backend != "bsdmm"),
        "tail_ms": 1e3 * dt / args.steps - (2 if backend == "bsdmm" else 1) * k1_avg_ms,     # the step minus its K1 launches
        "tail_note": "step minus K1: the update kernel(s) of the back-end (adaprox: k_ada_tail) and, in mode f16x2r at K1's K = 64 / 128, the launches of the "
                     "K x K correction (k_gfix_*)",
    }
    info = dev.k1_info()
    if info["chain"]:
        out["roofline"]["kernel"] += "<chain %d>" % info["chain"]     # gA summed in place along workgroup chains
    out["roofline"]["k1_layout"] = info
    if not args.rows:
        tr = pmc_traffic(args.config, effective_mode(dev))
        out["roofline"]["traffic"] = tr["bytes_per_launch"] if tr else None
        out["roofline"]["traffic_unit"] = "HBM bytes per K1 launch, rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of this build (%s); null: not measured for this build; algorithmic: %d" % (
            "fetch %d + write %d, %s" % (tr["fetch_bytes"], tr["write_bytes"], tr.get("when", "")) if tr else "profiles/k1_traffic.json has no entry for source hash %s" % kernel_source_hash(), M * N * 4)
    meas = side.get("measured")
    if meas and "error" not in meas:
        # the nominal peak stays the denominator of `frac` (the contract's roofline); beside it: the same bytes against what THIS box streams, and
        # K1 against the floor of its own job under this package's power cap
        rl = out["roofline"]
        rl["peak_nominal"] = rl["peak"]
        rl["peak_measured_read_gbs"] = meas["hbm_read_gbs"]
        rl["peak_measured_copy_gbs"] = meas["hbm_copy_gbs"]
        rl["peak_measured_fp16_mfma_tflops_random_data"] = meas["fp16_mfma_tflops_random"]
        rl["frac_of_measured_read_peak"] = (M * N * 4 / (k1_avg_ms * 1e-3) / 1e9) / meas["hbm_read_gbs"] if rl.get("bound") == "hbm" else None
        if args.config == "cfg3" and M == 16384 and N == 16384 and effective_mode(dev) == "f16x2r":
            rl["energy_floor_ms"] = meas["energy_floor_ms"]
            rl["frac_of_energy_floor"] = meas["energy_floor_ms"] / k1_avg_ms
            rl["energy_floor_with_epilogue_ms"] = meas["energy_floor_with_epilogue_ms"]
            rl["energy_floor_note"] = ("K1's skeleton (its pass over Y with K1's own 8-byte requests + its 28 fp16 MFMAs per SIMD and block on operands that toggle like data, a barrier per "
                                       "block, nothing else) on this GPU under its power cap: frac_of_energy_floor = floor / K1's launch time")
    out.update({k: v for k, v in side.items() if k != "measured"})
    if meas:
        out["measured_on_this_box"] = meas
    if not args.no_cpu:
        dev.close()
        Yh = Y.cpu().numpy()
        del Y
        torch.cuda.empty_cache()
        if full and not args.skip_cpu_baseline:
            out["end_to_end"] = end_to_end_leg(Yh, A0, S0, unity)
            # (BEHIND the headline: two seconds of fp64 MFMA work in front of it would hand the timed region a package at another temperature)
            out["fp64_inputs"] = fp64_inputs_leg(Yh, A0, S0, backend, unity, local, (meas or {}).get("fp64_mfma_tflops_random"))
        # the CPU leg is timed over the same steady state as the GPU: the last 5 warm-up iterations [warm_total - 5, warm_total)
        # of the GPU run against the same iteration indices of the oracle (its transient runs untimed)
        n_cpu = 5 if M * N <= 16384 * 16384 else 3
        trans = max(warm_total - n_cpu, 1) if backend == "adaprox" else 1
        rec, cpu_sub = (None, None) if args.skip_cpu_baseline else cpu_baseline(Yh, A0, S0, backend, unity, n_iter=n_cpu, transient=trans)
        if rec is not None:
            out["cpu_baseline"] = rec
        if rec is not None and backend == "adaprox" and cpu_sub is not None:
            rec["sub_iterations_per_step"] = cpu_sub
            rec["gpu_sub_iterations_same_window"] = sub_window
            rec["sub_iterations_equal"] = bool(sub_window is not None and all(abs(a - b) < 1e-9 for a, b in zip(cpu_sub, sub_window)))
    emit(out)


def _cfg3_call(Yany, A, S, unity, n):
    """one call of the PUBLIC entry point on cfg3's workload: `import proxmin_amd as proxmin; proxmin.nmf.nmf(...)`, library defaults"""
    from functools import partial
    import proxmin_amd as proxmin
    t0 = time.perf_counter()
    proxmin.nmf.nmf(Yany, A, S, prox_A=proxmin.operators.prox_plus,
                    prox_S=partial(proxmin.operators.prox_unity_plus, axis=0) if unity else proxmin.operators.prox_plus,
                    algorithm=proxmin.algorithms.adaprox, scheme="amsgrad", max_iter=n, e_rel=1e-3, check_convergence=False)
    return time.perf_counter() - t0


def nmf_call_leg(Yd, A0, S0, unity):
    """[r6] The drop-in path itself: nmf() with Y ALREADY IN HBM (a torch tensor: adopted in place, engine.DeviceArrayRef), host factors, no
    set_default_mode(), no DeviceNMF in sight.  Two calls (n1 and n2 iterations from the same start); the marginal rate (n2 - n1) / (t2 - t1)
    is the iteration rate a user of the public entry point gets, free of the per-call fixed costs, which are printed as well."""
    try:
        n1, n2 = 60, 460
        t1 = _cfg3_call(Yd, A0.copy(), S0.copy(), unity, n1)
        t2 = _cfg3_call(Yd, A0.copy(), S0.copy(), unity, n2)
        return {"iterations": [n1, n2], "seconds": [t1, t2], "value": (n2 - n1) / (t2 - t1), "unit": "it/s", "fixed_ms_per_call": 1e3 * (t1 - n1 * (t2 - t1) / (n2 - n1)),
                "note": "proxmin_amd.nmf.nmf(Y_on_gpu, A, S, algorithm=adaprox, scheme='amsgrad', prox_S=partial(prox_unity_plus, axis=0), check_convergence=False) in the library's "
                        "default mode: marginal iterations/s between a %d- and a %d-iteration call (iterations %d..%d of the problem; the headline times iterations ~60..160: K1's time "
                        "drifts a few per cent along a run, DESIGN.md section 5); fixed part = context, factor upload / download, the cold start's proximal passes" % (n1, n2, n1, n2)}
    except Exception as exc:
        return {"error": repr(exc)}


def fp64_inputs_leg(Yh32, A0, S0, backend, unity, local, peak_tf):
    """[r6] The reference computes in the dtype of its arrays (nmf.py:39-41) and its own examples are fp64.  The same workload handed over as
    float64 arrays runs the fp64 kernels (PMX_MODE_F64 at size: proxmin_amd/csrc/k_grad_f64.hip -- one v_mfma_f64_16x16x4_f64 pass per
    gradient, 8 M N K FLOPs per iteration -- and k_big_f64.hip): iterations/s, and the fraction of the fp64 matrix-core rate MEASURED on this
    box just before (measured_on_this_box).  MFMA-bound, not HBM-bound: Y (2 GiB in fp64) is streamed twice per iteration at ~1.5 TB/s each."""
    try:
        from proxmin_amd.engine import DeviceNMF
        M, N = Yh32.shape
        K = A0.shape[1]
        Yh = Yh32.astype(np.float64)
        with DeviceNMF(M, N, K, device=local, mode="f64") as dev:
            dev.set_Y(Yh)
            del Yh
            dev.set_factors(A0.astype(np.float64), S0.astype(np.float64))
            kernel = dev.k1_info()["kernel"]
            run = begin_solver(dev, backend, unity)
            warm, steps = 30, 20
            run(warm)
            t0 = time.perf_counter()
            run(steps)
            dt = time.perf_counter() - t0
        flops = 8.0 * M * N * K
        out = {"value": steps / dt, "unit": "it/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warm, "dtype": "f64", "kernel": kernel,
               "k1_tflops_achieved_whole_iteration": flops / (dt / steps) / 1e12,
               "note": "the headline's workload and solver with float64 arrays (Y uploaded from the host: 2 GiB): fp64 operands, products and sums; K1 = two MFMA passes "
                       "(8 M N K FLOPs), the rate is the whole iteration's"}
        if peak_tf:
            out["peak_tflops_measured"] = peak_tf
            out["frac_of_measured_fp64_mfma_peak"] = out["k1_tflops_achieved_whole_iteration"] / peak_tf
        return out
    except Exception as exc:
        return {"error": repr(exc)}


def end_to_end_leg(Yh, A0, S0, unity):
    """"D2H of A, S at exit included only in end-to-end nmf() latency, reported separately": nmf() called with HOST arrays (the
    reference's own calling convention: Y in host RAM), 200 iterations: upload of the 1 GiB Y through the C ABI, context, solver, write-back.
    PCIe-inclusive; never part of `value`."""
    try:
        n = 200
        t = _cfg3_call(Yh, A0.copy(), S0.copy(), unity, n)
        return {"iterations": n, "end_to_end_ms": 1e3 * t, "its_including_upload": n / t,
                "note": "nmf() from host NumPy arrays (Y 1 GiB pageable memory -> HBM, factors back at exit), %d iterations, one call" % n}
    except Exception as exc:
        return {"error": repr(exc)}


if __name__ == "__main__":
    main()
