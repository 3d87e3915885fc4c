"""bench.py's multi-GPU leg (`python bench.py --gpus N`, launched by torch.distributed.run with one rank per GPU): the rows of
Y / A sharded over the ranks, gS reduced once per iteration (proxmin_amd/distributed.py holds the protocol; this file only
builds the synthetic problem, times the loop the way the bench contract asks and prints what was measured).  Kept at the
repository root next to bench.py: measurement code, not part of the product package."""
import os
import time

import numpy as np

from proxmin_amd.distributed import (OneRankOfMany, ShardEngine, ShardedAdaproxDriver, ShardedLoop, _collectives, shard_rows)


def bench_sharded(args, M, N, K, backend, unity, desc, rank, world, local):
    """bench.py leg for --gpus N > 1: rows of Y / A sharded over the ranks (strong scaling)."""
    import torch
    import torch.distributed as dist
    from functools import partial
    from proxmin_amd import operators as ops
    from proxmin_amd.engine import DeviceNMF

    # PMX_DIST_BACKEND / PMX_BENCH_DEVICE: test-only overrides (two ranks on one GPU over gloo; RCCL needs a GPU per rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")      # (PMX_FORCE_SHARDED=1 without a launcher: a world of one)
    os.environ.setdefault("MASTER_PORT", "29531")
    if not dist.is_initialized():
        dist.init_process_group(backend=os.environ.get("PMX_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if "PMX_BENCH_DEVICE" in os.environ:
        local = int(os.environ["PMX_BENCH_DEVICE"])
        torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    r0, r1 = shard_rows(M, world)[rank]
    Ml = r1 - r0
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    At = torch.rand((M, K), generator=g, device=device, dtype=torch.float32)
    St = torch.rand((K, N), generator=g, device=device, dtype=torch.float32)
    if unity:
        St /= St.sum(0, keepdim=True)
    g.manual_seed(1234 + 7919 * (rank + 1))
    Y = At[r0:r1] @ St
    Y += 0.01 * torch.randn((Ml, N), generator=g, device=device, dtype=torch.float32)
    del At, St
    rng = np.random.default_rng(1234)
    A0 = rng.random((M, K), dtype=np.float32)[r0:r1].copy()
    S0 = rng.random((K, N), dtype=np.float32)
    if unity:
        S0 /= S0.sum(0, keepdims=True)
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(tstream)          # collectives order against the current stream
    own = os.environ.get("PMX_BENCH_OWN_STREAM", "0") == "1"     # tuning only (one process playing one rank): the library's own stream, collectives unordered
    dev = DeviceNMF(Ml, N, K, device=local, stream=None if own else tstream.cuda_stream, mode=getattr(args, "mode", None))
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    # PMX_BENCH_FAKE_WORLD=W (single process): this GPU plays rank 0 of W -- M is the rank's share, the collectives are local
    fake = int(os.environ.get("PMX_BENCH_FAKE_WORLD", "0")) if world == 1 else 0
    eff_world = fake if fake > 1 else world
    # S-split whenever it applies: adaprox with a projection-type prox_S (cfg4), N divisible by the rank count
    s_split = ((backend == "adaprox" and not unity) or backend == "pgm") and eff_world > 1 and N % eff_world == 0 and os.environ.get("PMX_S_SPLIT", "1") != "0"
    eng = ShardEngine(dev, eff_world, 0 if fake > 1 else rank, M * fake if fake > 1 else M, backend, s_split=s_split)
    pA = ops.device_proxseq(ops.prox_plus, 0)
    pS = ops.device_proxseq(partial(ops.prox_unity_plus, axis=0) if unity else ops.prox_plus, 1)
    # adaprox: the untimed warm-up continues until the proximal loops are past their start-up transient (as in the
    # single-GPU leg of bench.py), at least 20 iterations
    warm = max(args.warmup, 20) if backend == "adaprox" else args.warmup
    total = warm + args.steps
    # the collectives: local stand-ins when one process plays rank 0 of W, else torch.distributed or ($PMX_COMM=native) RCCL
    # through the C ABI
    from proxmin_amd.distributed import default_comm
    comm_default = default_comm(dist.get_backend() if dist.is_initialized() else "nccl")
    coll = OneRankOfMany() if fake > 1 else _collectives(None, dev, rank, world, None)
    if backend == "adaprox":
        dev.adaprox_begin([pA, pS], scheme="amsgrad", check_convergence=False, prox_max_iter=1000, e_rel=(1e-3, 1e-3))
        drv = ShardedAdaproxDriver(eng, None, False, True, 1000, chunk=int(os.environ["PMX_BENCH_CHUNK"]) if "PMX_BENCH_CHUNK" in os.environ else None, dist_module=coll)
        b1 = np.full(total, 0.9)
        run = lambda n: drv.run(n, b1)
    elif backend == "pgm":
        dev.pgm_begin([pA, pS], accelerated=False, e_rel=(1e-12, 1e-12))
        eng.bind_eval_buffer()
        loop = ShardedLoop(eng, None, deferred_test=True, dist_module=coll)
        run = loop.run
    else:
        pg = [ops.device_proxseq(ops.prox_plus, 0), ops.device_proxseq(partial(ops.prox_soft, thresh=1e-3), 0)]
        dev.bsdmm_begin([pA, pS], [pg, pg], e_rel=(1e-12, 1e-12), e_abs=(0.0, 0.0))
        loop = ShardedLoop(eng, None, deferred_test=False, dist_module=coll)
        run = loop.run
    run(warm)
    dev.set_timing(True, every=4)   # HIP events around every 4th K1 launch of the timed region
    if os.environ.get("PMX_BENCH_NO_PHASES", "0") != "1":
        dev.set_phase_timing(4)     # ... and at the phase boundaries of the same iterations (K1 / pack / collective / post / update)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    t_enq = time.perf_counter()     # the host is done enqueueing (it waited for the device once per chunk of iterations)
    torch.cuda.synchronize()
    dist.barrier()
    t1 = time.perf_counter()
    k1_ms, k1_n = dev.get_timing()
    phases, phases_n = dev.get_phase_timing()
    dev.set_phase_timing(0)
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    k1 = torch.tensor([k1_ms / max(k1_n, 1)], dtype=torch.float64, device=device)
    dist.all_reduce(k1, op=dist.ReduceOp.MAX)
    k1_avg_ms = float(k1.item())
    nk1 = 2 if backend == "bsdmm" else 1            # bsdmm: two K1 launches of 4 MNK each per iteration
    flop_per_it = (8.0 if backend == "bsdmm" else 6.0) * M * N * K
    info = dev.k1_info()
    eff_mode = "f32" if info["kernel"] in ("k_grad_f32", "k_grad_f32_pc") else dev.mode      # a split mode falls back to fp32 where it has no kernel
    its = args.steps / dt
    ach = (flop_per_it / nk1 * Ml / M) / (k1_avg_ms * 1e-3) / 1e12
    out = {
        "metric": "NMF iterations/sec at Y=%dx%d, K=%d" % (M, N, K),
        "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_effective": warm,
        "ms_per_step": 1e3 * dt / args.steps, "host_enqueue_done_ms_per_step": 1e3 * (t_enq - t0) / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": getattr(args, "mode_dtype", {}).get(eff_mode, eff_mode), "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.config, desc), "M": M, "N": N, "K": K, "backend": backend,
                   "mode": getattr(args, "mode_desc", {}).get(eff_mode, eff_mode),
                   "parallelism": ("rows of Y/A sharded over %d GPUs; S update sharded as well: one RCCL reduce-scatter of gS (%d floats) + one all-gather of S per iteration" if s_split else
                                   "rows of Y/A sharded over %d GPUs, one RCCL all-reduce of gS (%d floats) per iteration") % (eff_world, eng.layout.count)
                                  + ("; ONE process playing rank 0 of %d (local collectives)" % fake if fake > 1 else "")},
        "gflops": flop_per_it * its / 1e9,
        # rank 0's own iteration, phase by phase (HIP events on its stream, every 4th iteration of the timed region): where
        # the whole-job time goes -- "collective" is everything between the end of pack and the first kernel behind the
        # collective (waiting for slower ranks included), "after_update_to_next_k1" holds the S-split's all-gather
        "phases_ms": dict(phases, iterations_averaged=phases_n, sum=sum(phases.values()),
                          collectives=("native RCCL through the C ABI (pmx_comm_*)" if os.environ.get("PMX_COMM", comm_default) == "native" and not fake > 1
                                       else ("local stand-ins (one process playing rank 0)" if fake > 1 else "torch.distributed"))),
        "roofline": ({"kernel": info["kernel"], "bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s",
                      "frac": ach / 157.3, "traffic": None, "avg_launch_ms": k1_avg_ms, "launches": k1_n,
                      "k1_share_of_step": k1_avg_ms * nk1 * args.steps / (1e3 * dt), "k1_layout": info} if eff_mode == "f32" else
                     {"kernel": "k_grad_f16_k128", "bound": "mfma", "achieved": 3.0 * ach, "peak": 2500.0, "unit": "TFLOP/s",
                      "frac": 3.0 * ach / 2500.0, "traffic": None, "avg_launch_ms": k1_avg_ms, "launches": k1_n, "algorithmic_tflops": ach,
                      "k1_share_of_step": k1_avg_ms * nk1 * args.steps / (1e3 * dt), "k1_layout": info,
                      "note": "achieved = issued fp16 MFMA flops (3 products per fp32-class MAC)"} if info["kernel"] == "k_grad_f16_k128" else
                     {"kernel": (info["kernel"].replace("_r3", "<R3>") + ("<chain %d>" % info["chain"] if info["chain"] else "")) if info["kernel"] != "k_grad_bf16" or not (K == 64 and Ml % 128 == 0 and N % 256 == 0)
                      else "k_grad_bf16_v7" + ("<chain %d>" % info["chain"] if info["chain"] else ""), "bound": "hbm", "k1_layout": info,     # (pmx_k1_info names the kernel that ran)
                      "achieved": Ml * N * 4 / (k1_avg_ms * 1e-3) / 1e9, "peak": 8000.0,
                      "unit": "GB/s", "frac": Ml * N * 4 / (k1_avg_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                      "avg_launch_ms": k1_avg_ms, "launches": k1_n, "algorithmic_tflops": ach,
                      "k1_share_of_step": k1_avg_ms * nk1 * args.steps / (1e3 * dt)}),
    }
    dev.close()
    dist.destroy_process_group()
    return out
