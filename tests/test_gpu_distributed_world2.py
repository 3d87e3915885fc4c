"""GPU, world_size 2: TWO processes on the ONE visible GPU, gloo backend (it reduces CUDA tensors through the host;
RCCL refuses two ranks on one device).  This runs the real HIP phase entry points with rank != 0 and world > 1 --
row offsets, the global step rules from the all-reduced Gram matrix / column sums, the deferred stopping test --
and checks every rank against the single-GPU nmf() of the whole problem AND [r4] against the fp64 oracle run on the whole
problem (the single-GPU run is this library too: agreeing with it proves the protocol, not the arithmetic).  The 8-GPU RCCL
runs are the driver's."""
import os
import socket
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "adaprox_unity": dict(M=768, N=900, K=24, unity=True, its=9),
    "adaprox_k64_blocks": dict(M=1024, N=1280, K=64, unity=False, its=6),      # rows per rank 512, N % 256 = 0: the fast 16-bit-split kernels / the f32 whole-block kernel
    "adaprox_k64_r3": dict(M=1024, N=1280, K=64, unity=True, its=6, modes=("f16x2r",)),      # [r4] k_grad_f16_v8<R3> on both ranks (mode f16x2r)
    "adaprox_k128": dict(M=512, N=640, K=128, unity=False, its=5, modes=("f32", "f16x2")),   # 256 rows per rank: k_grad_f16_k128 in mode f16x2
    # S-split: the S update sharded too (reduce-scatter -> each rank updates N / 2 columns and their moments -> all-gather);
    # the cases above keep S replicated behind an all-reduce (s_split=False), these run the other mode on the same problems
    "adaprox_k64_split": dict(M=1024, N=1280, K=64, unity=False, its=6, s_split=True),
    "adaprox_k128_split": dict(M=512, N=640, K=128, unity=False, its=5, modes=("f32", "f16x2"), s_split=True),
    "pgm": dict(M=520, N=700, K=12, its=7),
    # [r4] S-split for pgm / FISTA: A's partial Gram matrix rides in the reduce-scatter's chunks, the ranks gather the next
    # evaluation point (the extrapolated iterate under FISTA) and, once, the iterate itself
    "pgm_split": dict(M=520, N=700, K=12, its=7, s_split=True),
    "fista_split": dict(M=1024, N=1280, K=64, its=7, s_split=True, accelerated=True),
    "bsdmm": dict(M=480, N=640, K=10, its=6),
    # [r4] ragged shards: 1000 rows per rank x 1500 columns run the tuned kernels on a zero-padded 1024 x 1536 frame (pmx_k1_frame);
    # the pack / post kernels and the collectives see the real N
    "adaprox_ragged": dict(M=2000, N=1500, K=64, unity=False, its=6, scheme="adam"),    # (adam: no eps clamp, every entry is held to the bound)
    "pgm_ragged_split": dict(M=2000, N=1500, K=64, its=6, s_split=True),
    # 2048 rows per rank x 16384: the chained K1 (chains of 4 workgroups).  Rank 1's third chained launch reports a fault
    # (PMX_INJECT_K1_FAULT): it falls back to slabs, rank 0 is stopped at the same iteration through the collective halt
    # flag, both go on from there.  (Two processes on one GPU can also fault for real -- not co-resident -- same path.)
    # [r4] the range guard of the two-term fp16 kernels on ONE rank: the rows of rank 1 start 1e5 x above the data, its first K1 launch is
    # refused (K max|A_local| max|S| > 2^16 max|Y_local|), it goes on in exact fp32; rank 0 -- whose rows are ordinary -- is stopped at the
    # same iteration through the collective halt flag, repeats it and stays on the fp16 kernel (tests/test_gpu_range.py; adam: no eps clamp)
    "adaprox_range_fault": dict(M=2048, N=1024, K=64, unity=False, its=5, scheme="adam", modes=("f16x2",), far_rows=(1024, 2048, 1e5)),
    "adaprox_chain_fault": dict(M=4096, N=16384, K=64, unity=True, its=6, modes=("f16x2", "f32", "bf16x3"), inject={1: "3"}),   # (k_grad_f16_v8 / k_grad_f32_pc / k_grad_bf16_v7)
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


JOIN_S = int(os.environ.get("PMX_W2_TIMEOUT", "300"))      # seconds a rank may take before the test gives up on it


def _worker(rank, world, port, name, mode, out_dir):
    # a rank that is still running shortly before the parent gives up writes where every thread stands (read back by the test)
    import faulthandler
    _fh = open(os.path.join(out_dir, "stuck_rank%d.txt" % rank), "w")
    faulthandler.dump_traceback_later(max(JOIN_S - 15, 5), file=_fh, exit=False)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # two processes share the GPU here: the fused adaprox tail (the product configuration's) needs all of it to itself, so
    # the ranks' tails take turns under an inter-process lock (test-only: one process per GPU never sets it)
    os.environ["PMX_TAIL_LOCKFILE"] = os.path.join(out_dir, "tail.lock")
    if CASES[name].get("inject"):
        # the chained-K1 fault case: two processes' chained K1s on ONE GPU are not co-resident either, a member may wait its
        # full 20 ms for a predecessor while it holds its CU -- ten times the tail's census patience.  That fight is the
        # point of the case (K1 falls back to slabs); the tail runs as separate kernels here.
        os.environ["PMX_TAIL_FUSED"] = "0"
    if rank in CASES[name].get("inject", {}):
        os.environ["PMX_INJECT_K1_FAULT"] = CASES[name]["inject"][rank]
    import torch
    import torch.distributed as dist
    import proxmin_amd as pm
    from proxmin_amd import distributed as pdist
    from oracle import nmf_oracle as orc
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pm.set_default_mode(mode)
        c = CASES[name]
        M, N, K = c["M"], c["N"], c["K"]
        Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=c.get("unity", False), seed=4)
        if c.get("far_rows"):
            A0[c["far_rows"][0]:c["far_rows"][1]] *= np.float32(c["far_rows"][2])
        r0, r1 = pdist.shard_rows(M, world)[rank]
        A_l, S = A0[r0:r1].copy(), S0.copy()
        ops = pm.operators
        # [r6] some cases enter through the PUBLIC front door -- nmf(Y_local, A_local, S, ..., M_global=M) dispatches to the same drivers
        front = name in ("pgm", "bsdmm", "adaprox_k64_split", "fista_split")
        if front and name.startswith("adaprox"):
            ret = pm.nmf.nmf(Y[r0:r1], A_l, S, algorithm=pm.adaprox, scheme=c.get("scheme", "amsgrad"), max_iter=c["its"], e_rel=1e-3,
                             check_convergence=False, M_global=M, s_split=c.get("s_split", False))
            assert len(ret) == 4 and ret[0] == (None, None)
            n = c["its"]
        elif front and name.startswith(("pgm", "fista")):
            kwp = dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)) if c.get("accelerated") else {}
            ret = pm.nmf.nmf(Y[r0:r1], A_l, S, max_iter=c["its"], e_rel=1e-9, M_global=M, s_split=c.get("s_split", False), **kwp)
            assert len(ret) == 3
            n = c["its"]
        elif front:
            pgl = [[ops.prox_plus, partial(ops.prox_soft, thresh=0.01)]] * 2
            ret = pm.nmf.nmf(Y[r0:r1], A_l, S, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=c["its"], e_rel=1e-9, M_global=M)
            assert len(ret) == 2
            n = c["its"]
        elif name.startswith("adaprox"):
            pS = partial(ops.prox_unity_plus, axis=0) if c.get("unity") else ops.prox_plus
            conv, n = pdist.nmf_adaprox_sharded(Y[r0:r1], A_l, S, M, prox_A=ops.prox_plus, prox_S=pS, scheme=c.get("scheme", "amsgrad"),
                                                check_convergence=False, e_rel=1e-3, max_iter=c["its"], s_split=c.get("s_split", False))
        elif name.startswith(("pgm", "fista")):
            conv, n = pdist.nmf_pgm_sharded(Y[r0:r1], A_l, S, M, e_rel=1e-9, max_iter=c["its"], s_split=c.get("s_split", False),
                                            accelerated=c.get("accelerated", False), step_scale=0.5 if c.get("accelerated") else 1.0)
        else:
            pgl = [[ops.prox_plus, partial(ops.prox_soft, thresh=0.01)]] * 2
            conv, n = pdist.nmf_bsdmm_sharded(Y[r0:r1], A_l, S, M, proxs_g=pgl, e_rel=1e-9, max_iter=c["its"])
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=A_l, S=S, n=n, r0=r0, r1=r1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2", "f16x2r"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_two_ranks_on_one_gpu_match_single_gpu(tmp_path, name, mode):
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    g.build()
    import proxmin_amd as pm
    from oracle import nmf_oracle as orc
    c = CASES[name]
    if mode not in c.get("modes", (mode,)) or (mode == "f16x2r" and "f16x2r" not in c.get("modes", ())):
        pytest.skip("case is specific to another arithmetic mode")
    M, N, K = c["M"], c["N"], c["K"]
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=c.get("unity", False), seed=4)
    if c.get("far_rows"):
        A0[c["far_rows"][0]:c["far_rows"][1]] *= np.float32(c["far_rows"][2])
    ops = pm.operators
    pm.set_default_mode(mode)
    try:
        A1, S1 = A0.copy(), S0.copy()
        tb = pm.utils.Traceback()
        if name.startswith("adaprox"):
            pS = partial(ops.prox_unity_plus, axis=0) if c.get("unity") else ops.prox_plus
            pm.nmf.nmf(Y, A1, S1, algorithm=pm.adaprox, scheme=c.get("scheme", "amsgrad"), prox_S=pS, max_iter=c["its"], e_rel=1e-3,
                       check_convergence=False, callback=tb)
        elif name.startswith(("pgm", "fista")):
            kwp = dict(accelerated=True, step=pm.nmf.scaled_step_pgm(0.5)) if c.get("accelerated") else {}
            pm.nmf.nmf(Y, A1, S1, max_iter=c["its"], e_rel=1e-9, callback=tb, **kwp)
        else:
            pgl = [[ops.prox_plus, partial(ops.prox_soft, thresh=0.01)]] * 2
            pm.nmf.nmf(Y, A1, S1, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=c["its"], e_rel=1e-9, callback=tb)
    finally:
        pm.set_default_mode(None)
    import time
    import warnings
    ctx = mp.get_context("spawn")
    for attempt in range(2):
        # Two processes on ONE GPU with gloo is a test-only arrangement, and once in several hundred executions (round 4: 1 of
        # ~500) a pair of ranks did not come back within the patience below.  A TIMEOUT (never a wrong result or a failed
        # rank) is retried once, with where the ranks stood written into the test's warnings.
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, name, mode, str(tmp_path))) for r in range(2)]
        for p in procs:
            p.start()
        deadline = time.time() + JOIN_S
        for p in procs:
            p.join(max(deadline - time.time(), 1))
        alive = [p for p in procs if p.is_alive()]
        for p in alive:                      # never leave a rank behind: it would keep the GPU (and pytest) busy
            p.terminate()
            p.join(10)
        if not alive:
            break
        stuck = ""
        for r in range(2):
            f = tmp_path / ("stuck_rank%d.txt" % r)
            if f.exists():
                stuck += "\n--- rank %d ---\n%s" % (r, f.read_text()[-3000:])
        msg = "rank process(es) did not finish within %d s (attempt %d)%s" % (JOIN_S, attempt + 1, stuck)
        if attempt == 1:
            raise AssertionError(msg)
        warnings.warn(msg)
    for p in procs:
        assert p.exitcode == 0, "rank process failed (exit code %r)" % p.exitcode
    S_ranks = []
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert int(z["n"]) == len(tb.trace)
        # the all-reduce sums the ranks' gS in a different order than the single-GPU slab fold: fp32 rounding only
        if c.get("inject") or c.get("far_rows"):
            # (far_rows: rank 0 stays on the fp16 kernel, the single-GPU run -- whose maxima are rank 1's -- is all fp32: two arithmetics)
            # after the fall-back this rank sums gA over slabs, the single-GPU run along chains: AMSGrad's eps clamp
            # turns that rounding difference into a visible one on a few entries per ten thousand (test_gpu_nmf.py)
            for got, want in ((z["A"], A1[int(z["r0"]):int(z["r1"])]), (z["S"], S1)):
                err = np.abs(got.astype(np.float64) - want)
                assert (err <= 1e-5 + 1e-4 * np.abs(want)).mean() >= 0.999
                # (the hard envelope is the eps-clamp's: one entry in a million at 265 x the bound was measured when k_grad_f32_pc's
                # gSt summation order changed in round 4; the fp32 oracle itself is up to 62 x away from the fp64 one at full cfg3)
                np.testing.assert_allclose(got, want, rtol=0.1, atol=0.01)
        else:
            np.testing.assert_allclose(z["A"], A1[int(z["r0"]):int(z["r1"])], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(z["S"], S1, rtol=1e-4, atol=1e-5)
        S_ranks.append(z["S"])
    np.testing.assert_array_equal(S_ranks[0], S_ranks[1])      # replicated state stays bit-identical across ranks
    # ---- every rank against the ORACLE (fp64, whole problem, identical fp32 inputs) --------------------------------------
    Ao, So, Y64 = A0.astype(np.float64), S0.astype(np.float64), Y.astype(np.float64)
    if name.startswith("adaprox"):
        orc.adaprox_nmf(Y64, Ao, So, ("plus",), ("unity_plus", 0) if c.get("unity") else ("plus",), scheme=c.get("scheme", "amsgrad"),
                        max_iter=c["its"], e_rel=1e-3, check_convergence=False)
    elif name.startswith(("pgm", "fista")):
        okw = dict(accelerated=True, step=lambda a, s_, it, g: tuple(0.5 * x for x in orc.lipschitz_steps(a, s_))) if c.get("accelerated") else {}
        orc.pgm_nmf(Y64, Ao, So, max_iter=c["its"], e_rel=1e-9, **okw)
    else:
        orc.bsdmm_nmf(Y64, Ao, So, proxs_g=[[("plus",), ("soft", 0.01, "relative")]] * 2, max_iter=c["its"], e_rel=1e-9)
    smooth = not name.startswith("adaprox") or c.get("scheme") == "adam"      # amsgrad's eps clamp: a fraction, as everywhere else (test_gpu_parity_strict.py)
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        for got, want, what in ((z["A"], Ao[int(z["r0"]):int(z["r1"])], "A rows of rank %d" % r), (z["S"], So, "S on rank %d" % r)):
            err = np.abs(got.astype(np.float64) - want)
            ratio = err / (1e-5 + 1e-4 * np.abs(want))
            if smooth and mode == "f32":
                assert ratio.max() <= 1.0, "%s: worst entry %.2f x the north star's bound against the fp64 oracle" % (what, ratio.max())
            else:
                assert (ratio <= 1.0).mean() >= (0.9999 if smooth else 0.995), "%s: %.5f within the bound" % (what, (ratio <= 1.0).mean())
                # hard envelope: 50 x the bound for the smooth back-ends in a split mode; amsgrad's eps clamp puts a few entries
                # per million further out against fp64 in ANY fp32 arithmetic (the oracle's own fp32 run: 62 x at full cfg3)
                env = 50.0 if smooth else 1000.0
                assert ratio.max() <= env, "%s: worst entry %.1f x the bound (envelope %g x)" % (what, ratio.max(), env)
