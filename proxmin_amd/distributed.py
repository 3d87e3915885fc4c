"""Row-sharded multi-GPU NMF (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The K-factor NMF shards naturally by rows: rank r owns rows [r0, r1) of Y and A
and their optimiser state; S and its state are replicated.  Everything is local except
    gS = sum_r A_r^T R_r          (nmf.py:41 decomposed by rows)
and three small sums (column sums of A for the adaprox step rule, and A's two stopping-test sums).
They travel in ONE float32 buffer that is all-reduced once per iteration:

    comm = [ gSt (N*K) | Gram(A) (KP*KP) | colsum(A) (128) | scalars (32) ]

The driver below is written against a small engine interface (phase / chain_status / more_subs /
comm) so that the protocol can be exercised on CPU with the gloo backend and a NumPy stand-in engine
(tests/test_distributed_cpu.py); the product engine is `ShardEngine`, a thin wrapper over the
libpmx phase entry points (HIP kernels).
"""
from __future__ import annotations

import ctypes as C
import time

import os

import numpy as np

from . import _lib

HALT_CONVERGED, HALT_NEED_SUB, HALT_ERROR, HALT_RETRY, HALT_PEER = 1, 2, 3, 4, 5
MAXK = 128
N_SCALARS = 32


def shard_rows(M, world):
    """Contiguous, nearly equal row ranges: returns list of (r0, r1)."""
    base, rem = divmod(M, world)
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((r0, r0 + n))
        r0 += n
    return out


class CommLayout:
    """Offsets (in floats) inside the all-reduce buffer; mirrors pmx_comm_layout."""

    def __init__(self, N, K):
        self.N, self.K = int(N), int(K)
        self.KP = 32 if K <= 32 else (64 if K <= 64 else 128)
        self.gram = self.N * self.K
        self.colsum = self.gram + self.KP * self.KP
        self.scalars = self.colsum + MAXK
        self.count = self.scalars + N_SCALARS


class SplitLayout:
    """S-split comm buffer: `world` chunks of [ gSt rows of rank q | Gram | colsum(A) | colsum(S) | scalars ]; mirrors
    pmx_comm_layout_split."""

    def __init__(self, N, K, world):
        assert N % world == 0
        self.N, self.K, self.world = int(N), int(K), int(world)
        self.KP = 32 if K <= 32 else (64 if K <= 64 else 128)
        self.sncol = self.N // self.world
        self.gram = self.sncol * self.K
        self.colsum_A = self.gram + self.KP * self.KP
        self.colsum_S = self.colsum_A + MAXK
        self.scalars = self.colsum_S + MAXK
        self.chunk = self.scalars + N_SCALARS
        self.count = self.chunk * self.world


def reduce_scatter_sum(dist, out, inp, group=None):
    """out (chunk) <- sum over ranks of chunk `rank` of inp.  RCCL: one reduce_scatter; back-ends without it (gloo: CPU
    tests, two ranks on one GPU) all-reduce the whole buffer and keep their chunk -- same result, test-only traffic."""
    try:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)
    except (RuntimeError, NotImplementedError):
        dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=group)
        r, n = dist.get_rank(group), out.numel()
        out.copy_(inp[r * n:(r + 1) * n])


def all_gather_chunks(dist, full, group=None):
    """every rank's chunk of `full` (equal contiguous pieces, rank order) to every rank, in place"""
    r, w = dist.get_rank(group), dist.get_world_size(group)
    n = full.numel() // w
    flat = full.view(-1)
    try:
        dist.all_gather_into_tensor(flat, flat[r * n:(r + 1) * n].clone() if flat.device.type == "cpu" else flat[r * n:(r + 1) * n], group=group)
    except (RuntimeError, NotImplementedError):
        parts = [flat[q * n:(q + 1) * n].clone() for q in range(w)]
        dist.all_gather(parts, parts[r], group=group)
        for q in range(w):
            if q != r:
                flat[q * n:(q + 1) * n].copy_(parts[q])


class OneRankOfMany:
    """Stand-in for torch.distributed in a ONE-process measurement of one rank's share of a W-rank run (bench.py with
    PMX_BENCH_FAKE_WORLD=W on a single GPU): the collectives move this rank's own data only (sums over one rank), so that the
    kernels of the sharded code path -- chunked pack, post, the S update on N / W columns -- can be timed and profiled."""

    class ReduceOp:
        SUM = None

    def all_reduce(self, t, op=None, group=None):
        return None

    def reduce_scatter_tensor(self, out, inp, op=None, group=None):
        out.copy_(inp[:out.numel()])

    def all_gather_into_tensor(self, out, inp, group=None):
        return None

    def get_rank(self, group=None):
        return 0

    def get_world_size(self, group=None):
        return 1


class NativeRccl:
    """torch.distributed's three collectives of the sharded drivers, on RCCL through the C ABI instead (include/pmx.h:
    pmx_comm_*): one communicator per context, every call enqueued on the context's own stream -- no Python-side stream
    juggling, nothing of torch in the data path.  The drivers only need this interface (all_reduce, reduce_scatter_tensor,
    all_gather_into_tensor, get_rank, get_world_size, ReduceOp.SUM), so an instance goes where `torch.distributed` went.
    `bootstrap(dev, rank, world, broadcast)`: rank 0 draws the 128-byte id, `broadcast(bytes_or_None) -> bytes` hands it to
    every rank (torch.distributed's broadcast_object_list, MPI, a file ...), every rank joins.  Rank 0 takes part in the
    broadcast WHATEVER happened to it: an id it could not draw (RCCL not loadable) travels as a marker, every rank raises
    the same error without entering pmx_comm_init, and the caller's agreement step (see _collectives) is reached by all."""

    class ReduceOp:
        SUM = None

    def __init__(self, dev, rank, world):
        self.dev, self.rank, self.world = dev, int(rank), int(world)

    @classmethod
    def bootstrap(cls, dev, rank, world, broadcast):
        uid = C.create_string_buffer(128)
        err = None
        if rank == 0:
            try:
                _lib.check(dev.lib.pmx_comm_unique_id(uid))
            except Exception as exc:     # noqa: BLE001 -- the other ranks are waiting in the broadcast: tell them
                err = exc
        raw = broadcast((bytes(uid.raw) if err is None else b"") if rank == 0 else None)
        if err is not None:
            raise err
        if not raw:
            raise _lib.PmxError("native RCCL collectives: rank 0 could not draw a communicator id")
        assert len(raw) == 128
        _lib.check(dev.lib.pmx_comm_init(dev.h, raw, int(rank), int(world)))
        return cls(dev, rank, world)

    def all_reduce(self, t, op=None, group=None):
        _lib.check(self.dev.lib.pmx_comm_all_reduce(self.dev.h, C.c_void_p(t.data_ptr()), t.numel()))

    def reduce_scatter_tensor(self, out, inp, op=None, group=None):
        assert inp.numel() == out.numel() * self.world
        _lib.check(self.dev.lib.pmx_comm_reduce_scatter(self.dev.h, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), out.numel()))

    def all_gather_into_tensor(self, out, inp, group=None):
        assert out.numel() == inp.numel() * self.world
        _lib.check(self.dev.lib.pmx_comm_all_gather(self.dev.h, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), inp.numel()))

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world


def default_comm(backend):
    """Who issues the per-iteration collectives when the caller does not say: RCCL through the C ABI ("native": every call
    enqueued on the context's own stream by the library, no per-iteration stream bookkeeping in Python) whenever the process
    group itself runs on RCCL (backend "nccl": one GPU per rank, the library loadable by construction); torch.distributed for
    anything else (gloo: the CPU protocol tests and several ranks on one GPU, where RCCL refuses to run)."""
    return "native" if "nccl" in str(backend).lower() else "torch"


def _collectives(comm, dev, rank, world, group):
    """`comm`: "torch" (torch.distributed on the current stream) | "native" (RCCL through the C ABI, bootstrapped over the
    process group that is there anyway) | None: $PMX_COMM, else default_comm(backend of the group) -- [r4] native on RCCL."""
    _lib.require_torch()
    import torch.distributed as dist
    comm_arg = comm
    comm = comm or os.environ.get("PMX_COMM") or default_comm(dist.get_backend(group))
    if comm == "torch":
        return dist
    if comm != "native":
        raise ValueError("comm must be 'torch' or 'native'")
    explicit = comm_arg is not None or bool(os.environ.get("PMX_COMM"))

    def broadcast(raw):
        box = [raw]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return box[0]
    # Join the communicator and prove it on a known answer before the solver depends on it; the ranks then AGREE (over the
    # process group that is there anyway) on whether every one of them succeeded.  Asked for explicitly, a failure raises;
    # chosen by default, the run goes on with torch.distributed's collectives and says so.
    torch = _lib.require_torch()
    err, nat = None, None
    try:
        nat = NativeRccl.bootstrap(dev, rank, world, broadcast)
        probe = torch.full((256,), float(rank + 1), dtype=torch.float32, device=torch.device("cuda", dev.device))
        torch.cuda.synchronize(dev.device)
        nat.all_reduce(probe)
        dev.sync()
        if not bool((probe == float(world * (world + 1) // 2)).all().item()):
            err = _lib.PmxError("native RCCL all-reduce self-test: wrong sum")
    except Exception as exc:           # noqa: BLE001 -- whatever went wrong, the ranks must still agree below
        err = exc
    if world > 1:
        okf = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float32, device=torch.device("cuda", dev.device))
        dist.all_reduce(okf, op=dist.ReduceOp.MIN, group=group)
        all_ok = bool(okf.item() >= 1.0)
    else:
        all_ok = err is None
    if all_ok:
        return nat
    if explicit:
        raise err if err is not None else _lib.PmxError("native RCCL collectives failed on another rank")
    import logging
    logging.getLogger("proxmin").warning("native RCCL collectives unavailable (%s): using torch.distributed's", err if err is not None else "failed on another rank")
    return dist


class ShardedAdaproxDriver:
    """Iteration loop of the row-sharded adaprox back-end (algorithms.py:365-413 with one all-reduce
    per iteration).  `engine` provides phase(), chain_status(), more_subs() and the `comm` tensor."""

    def __init__(self, engine, group=None, check_convergence=True, any_prox=True, prox_max_iter=1000, chunk=None, dist_module=None):
        if dist_module is None:
            _lib.require_torch()
            import torch.distributed as dist_module
        self.dist = dist_module
        self.eng = engine
        self.group = group
        self.check = bool(check_convergence)
        self.any_prox = bool(any_prox)
        self.prox_max_iter = int(prox_max_iter)
        if chunk is None:
            # iterations enqueued between two reads of the device status (each read drains the stream: 8 us per iteration at 16, 2 us at
            # 64, cfg4's share).  The fused tail decides its proximal loops on the device, so nothing is speculated on and the chunks may
            # be as long as the single-GPU loop's (pmx_adaprox_run); the chain of tail kernels keeps 16 (its per-iteration records: 64 slots)
            chunk = None
        self._chunk_arg = None if chunk is None else int(chunk)
        self.chunk = self._chunk()
        self.nsub = 2
        self.it = 0              # completed iterations
        self.stopped = False

    SUB_REC_SLOTS = 64       # per-iteration records of the chain of tail kernels (k_update.hip: sub_rec[it & 63])

    def _chunk(self):
        """[r6] Re-evaluated at every chunk boundary: the fused tail can fault MID-RUN (pmx_adaprox_phase then clears tail_fused and
        the chain of tail kernels takes over) -- a chunk of 64 fixed at construction would leave that chain with exactly as many iterations
        in flight as it has record slots.  The chain keeps 16; an explicit chunk is clamped strictly below the slot count while it runs."""
        fused = getattr(self.eng, "tail_fused", None)
        fused = bool(fused() if callable(fused) else fused)
        if self._chunk_arg is None:
            return 64 if fused else 16
        return self._chunk_arg if fused else min(self._chunk_arg, self.SUB_REC_SLOTS - 1)

    def _allreduce(self):
        if getattr(self.eng, "s_split", False):     # S-split: reduce-scatter here, the all-gather follows the update
            reduce_scatter_sum(self.dist, self.eng.comm_out, self.eng.comm, self.group)
        else:
            self.dist.all_reduce(self.eng.comm, op=self.dist.ReduceOp.SUM, group=self.group)

    def _iteration(self, it, b1):
        b1_prev = b1[it - 1]    # b1[-1] at it = 0, like the reference (algorithms.py:213)
        self.eng.phase(0, it, b1[it], b1_prev, 0)
        self._allreduce()
        self.eng.phase(1, it, b1[it], b1_prev, self.nsub)
        if getattr(self.eng, "s_split", False):
            all_gather_chunks(self.dist, self.eng.st_full, self.group)      # every rank's updated columns of S

    def run(self, n_iter, b1):
        """Advance up to n_iter iterations; b1 is the full per-iteration array (len >= it + n_iter)."""
        target = self.it + int(n_iter)
        while self.it < target and not self.stopped:
            self.chunk = self._chunk()
            hi = min(target, self.it + self.chunk)
            first = self.it
            for it in range(first, hi):
                self._iteration(it, b1)
            halted, reason, it_done, tau = self.eng.chain_status()
            t_enq = self.nsub
            while halted and reason == HALT_NEED_SUB:
                # iteration `it_done` ran out of proximal sub-iteration passes (same on every rank: the
                # S block is replicated and A's projection-type prox never needs more than two passes).
                # Under S-split every rank owns other columns of S: a rank-local shortage would re-enqueue collectives on
                # one rank only -- S-split is restricted to projection-type prox_S so that this cannot happen.
                if getattr(self.eng, "s_split", False):
                    raise _lib.PmxError("S-split: a rank-local proximal loop ran out of passes (HALT_NEED_SUB); projection-type "
                                        "operators never do -- refusing to re-enqueue collectives on one rank only")
                more = min(max(4, t_enq), 64)
                self.eng.more_subs(t_enq, more)
                t_enq += more
                for it in range(it_done + 1, hi):
                    self._iteration(it, b1)
                before = it_done
                halted, reason, it_done, tau = self.eng.chain_status()
                if it_done > before:
                    if self.any_prox:   # the iterations re-enqueued from here on start from the count the last one took
                        self.nsub = max(2, min(max(tau), self.prox_max_iter))
                    t_enq = self.nsub
            self.it = it_done
            if self.any_prox:
                self.nsub = max(2, min(max(tau), self.prox_max_iter))
            if halted and reason == HALT_CONVERGED:
                self.stopped = True
            elif halted and reason == HALT_ERROR:
                raise _lib.PmxError("the device chain of rank-local kernels reported an error")
            # HALT_RETRY (this rank fell back after a recoverable kernel fault) / HALT_PEER (another rank did): every rank
            # stopped before the update of iteration it_done (collective halt flag), the halt is cleared: go on from there
        if self.check and not self.stopped and n_iter > 0:
            # the stopping test of the last iteration has not been evaluated yet (it needs A's global sums)
            self.eng.phase(2, self.it, 0.0, 0.0, 0)
            self._allreduce()
            self.eng.phase(3, self.it, 0.0, 0.0, 0)
            halted, reason, it_done, tau = self.eng.chain_status()
            if halted and reason == HALT_CONVERGED:
                self.stopped = True
        return self.it


class ShardedLoop:
    """Iteration loop of the row-sharded pgm and bsdmm back-ends: phase 0, ONE all-reduce, phase 1.
    pgm evaluates its stopping test one iteration late (A's sums are global only after the next all-reduce)
    and flushes it after the last iteration; bsdmm's all-reduce sits between its A step and its S step, so its
    test is exact without deferral."""

    def __init__(self, engine, group=None, deferred_test=True, chunk=32, dist_module=None):
        # chunk: iterations enqueued between two reads of the device status.  After a mid-chunk convergence the rest of the chunk still runs
        # (halted kernels return at once, the collectives are real): 32 bounds that tail at half of round 5's 64 for ~1 us more per iteration
        if dist_module is None:
            _lib.require_torch()
            import torch.distributed as dist_module
        self.dist, self.eng, self.group = dist_module, engine, group
        self.deferred = bool(deferred_test)
        self.chunk = int(chunk)
        self.it = 0
        self.stopped = False

    def _allreduce(self):
        if getattr(self.eng, "s_split", False):     # S-split: reduce-scatter here, the all-gather follows the update
            reduce_scatter_sum(self.dist, self.eng.comm_out, self.eng.comm, self.group)
        else:
            self.dist.all_reduce(self.eng.comm, op=self.dist.ReduceOp.SUM, group=self.group)

    def run(self, n_iter):
        target = self.it + int(n_iter)
        split = getattr(self.eng, "s_split", False)
        while self.it < target and not self.stopped:
            hi = min(target, self.it + self.chunk)
            for it in range(self.it, hi):
                self.eng.phase(0, it)
                self._allreduce()
                self.eng.phase(1, it)
                if split:                            # every rank's columns of the next evaluation point
                    all_gather_chunks(self.dist, self.eng.st_full, self.group)
            halted, reason, it_done, _ = self.eng.chain_status()
            self.it = it_done
            if halted and reason == HALT_CONVERGED:
                self.stopped = True
            elif halted and reason == HALT_ERROR:
                raise _lib.PmxError("the device chain of rank-local kernels reported an error")
        if self.deferred and not self.stopped and n_iter > 0:
            self.eng.phase(2, self.it)
            self._allreduce()
            self.eng.phase(3, self.it)
            halted, reason, _, _ = self.eng.chain_status()
            self.stopped = bool(halted and reason == HALT_CONVERGED)
        if split and getattr(self.eng, "st_iterate", None) is not None and n_iter > 0:
            all_gather_chunks(self.dist, self.eng.st_iterate, self.group)      # FISTA: the iterate S itself, once per run
        return self.it


class ShardEngine:
    """libpmx-backed engine for one rank (HIP kernels; comm buffer is a torch CUDA tensor)."""

    def __init__(self, dev, world, rank, M_global, algorithm="adaprox", s_split=False):
        self.algorithm = algorithm
        torch = _lib.require_torch()
        self.dev = dev
        lib = dev.lib
        _lib.check(lib.pmx_set_world(dev.h, rank, world, int(M_global)))
        self.s_split = bool(s_split)
        if self.s_split:
            assert algorithm in ("adaprox", "pgm"), "S-split is implemented for the adaprox and pgm back-ends"
            _lib.check(lib.pmx_set_s_split(dev.h, 1))
            cnt, chunk = C.c_int64(), C.c_int64()
            offs = (C.c_int64 * 4)()
            _lib.check(lib.pmx_comm_layout_split(dev.h, C.byref(cnt), C.byref(chunk), offs))
            self.layout = SplitLayout(dev.N, dev.K, world)
            assert (cnt.value, chunk.value, offs[0], offs[1], offs[2], offs[3]) == (
                self.layout.count, self.layout.chunk, self.layout.gram, self.layout.colsum_A, self.layout.colsum_S, self.layout.scalars)
            device = torch.device("cuda", dev.device)
            self.comm = torch.zeros(cnt.value, dtype=torch.float32, device=device)
            self.comm_out = torch.zeros(chunk.value, dtype=torch.float32, device=device)
            _lib.check(lib.pmx_set_comm_buffer(dev.h, C.c_void_p(self.comm.data_ptr()), cnt.value))
            _lib.check(lib.pmx_set_comm_out(dev.h, C.c_void_p(self.comm_out.data_ptr()), chunk.value))
            self.st_full = self._alias(_lib.BUF_ST)      # S^T (N x K) as the library holds it: the all-gather buffer
            self.st_iterate = None                       # pgm / FISTA: see bind_eval_buffer()
            return
        cnt = C.c_int64()
        offs = (C.c_int64 * 3)()
        _lib.check(lib.pmx_comm_layout(dev.h, C.byref(cnt), offs))
        self.layout = CommLayout(dev.N, dev.K)
        assert (cnt.value, offs[0], offs[1], offs[2]) == (self.layout.count, self.layout.gram, self.layout.colsum, self.layout.scalars)
        self.comm = torch.zeros(cnt.value, dtype=torch.float32, device=torch.device("cuda", dev.device))
        _lib.check(lib.pmx_set_comm_buffer(dev.h, C.c_void_p(self.comm.data_ptr()), cnt.value))

    def _alias(self, buf):
        """zero-copy torch view of one of the library's N x K buffers (the all-gather writes into the library's own memory)"""
        ptr, n = C.c_void_p(), C.c_int64()
        _lib.check(self.dev.lib.pmx_buffer_ptr(self.dev.h, buf, C.byref(ptr), C.byref(n)))
        t = _device_tensor(ptr.value, n.value, self.dev.device)
        # a tensor that COPIED the buffer would leave K1 reading stale columns
        if t.data_ptr() != ptr.value or t.numel() != n.value:
            raise _lib.PmxError("S-split: torch did not alias the library's buffer %d (%#x, %d floats) but made a copy" % (buf, ptr.value, n.value))
        return t

    def bind_eval_buffer(self):
        """pgm, S-split, AFTER pgm_begin: what the ranks gather every iteration is the point the next gradient is evaluated at --
        the extrapolated iterate under FISTA (PMX_BUF_EVAL_ST), S itself otherwise; under FISTA the iterate S proper is gathered
        once, when a run ends (st_iterate)."""
        if not self.s_split or self.algorithm != "pgm":
            return
        ev = self._alias(_lib.BUF_EVAL_ST)
        if ev.data_ptr() != self.st_full.data_ptr():
            self.st_iterate, self.st_full = self.st_full, ev

    def tail_fused(self):
        """adaprox, after adaprox_begin: does the iteration tail run as ONE kernel (k_ada_tail)?"""
        return self.algorithm == "adaprox" and bool(self.dev.k1_info().get("tail_fused"))

    def phase(self, phase, it, b1_it=0.0, b1_prev=0.0, nsub=0):
        lib, h = self.dev.lib, self.dev.h
        if self.algorithm == "adaprox":
            _lib.check(lib.pmx_adaprox_phase(h, int(phase), int(it), float(b1_it), float(b1_prev), int(nsub)))
        elif self.algorithm == "pgm":
            _lib.check(lib.pmx_pgm_phase(h, int(phase), int(it)))
        else:
            _lib.check(lib.pmx_bsdmm_phase(h, int(phase)))

    def chain_status(self):
        h, r, i = C.c_int(), C.c_int(), C.c_int()
        tau = (C.c_int * 2)()
        _lib.check(self.dev.lib.pmx_chain_status(self.dev.h, C.byref(h), C.byref(r), C.byref(i), tau))
        return h.value, r.value, i.value, (tau[0], tau[1])

    def more_subs(self, t0, n):
        _lib.check(self.dev.lib.pmx_adaprox_more_subs(self.dev.h, int(t0), int(n)))


class _DevArray:
    """zero-copy view of a device buffer of the C library for torch (the __cuda_array_interface__ protocol)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}


def _device_tensor(ptr, n, device):
    torch = _lib.require_torch()
    return torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", device))


def projection_type(seq):
    """True if every operator of a device sequence is a coordinate-wise projection whose result does not depend on the
    proximal pass count (plus / id / zero / absolute min, max, hard): what S-split (and a sharded A) require."""
    box = {_lib.PROX[n] for n in ("id", "zero", "plus")}
    maybe = {_lib.PROX[n] for n in ("min", "max", "hard", "hard_plus") if n in _lib.PROX}
    return all(seq.seq[i].op in box or (seq.seq[i].op in maybe and not seq.seq[i].relative) for i in range(seq.n))


def nmf_adaprox_sharded(Y_local, A_local, S, M_global, prox_A=None, prox_S=None, scheme="adam", b1=0.9, b2=0.999,
                        eps=1e-8, p=0.25, check_convergence=True, e_rel=1e-3, max_iter=1000, prox_max_iter=1000,
                        group=None, device=None, Y_is_device_ptr=None, s_split="auto", comm=None):
    """Row-sharded counterpart of `nmf(Y, A, S, algorithm=adaprox, ...)` for one rank.

    Y_local: this rank's rows of Y (ndarray, M_local x N); A_local: the matching rows of A (updated in
    place); S: full K x N (replicated, updated in place, identical on every rank).
    Requires an initialised torch.distributed process group whose backend can reduce CUDA tensors
    (nccl = RCCL).  comm: "torch" (default) issues the collectives through torch.distributed, "native" through the C ABI's own
    RCCL entry points (pmx_comm_*; the process group only hands the communicator id around).  Returns (converged, iterations)."""
    torch = _lib.require_torch()
    _lib.require_torch()
    import torch.distributed as dist
    from . import operators
    from .engine import DeviceNMF

    if prox_A is None:
        prox_A = operators.prox_plus
    if prox_S is None:
        prox_S = operators.prox_plus
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None:
        device = torch.cuda.current_device()
    seqs = [operators.device_proxseq(prox_A, 0), operators.device_proxseq(prox_S, 1)]
    if not hasattr(b1, "__iter__"):
        b1 = np.array((b1,) * max_iter)
    b1 = np.asarray(b1, dtype=np.float64)
    e = (e_rel, e_rel) if np.isscalar(e_rel) else tuple(e_rel)
    # kernels and collectives must share a (non-default) stream: torch's collectives order themselves
    # against the CURRENT stream, so make ours current for the duration of the solve
    tstream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(tstream), DeviceNMF(A_local.shape[0], S.shape[1], A_local.shape[1], device=device,
                                               stream=tstream.cuda_stream) as dev:
        dev.set_Y(Y_local)
        dev.set_factors(A_local, S)
        # S-split ("auto": whenever it applies): the S update sharded over the ranks between a reduce-scatter and an
        # all-gather instead of replicated behind an all-reduce -- projection-type prox_S, N divisible by the rank count
        can_split = world > 1 and S.shape[1] % world == 0 and projection_type(seqs[1])
        if s_split is True and not can_split:
            raise NotImplementedError("S-split needs a projection-type prox_S and N divisible by the number of ranks")
        eng = ShardEngine(dev, world, rank, M_global, s_split=bool(s_split) and can_split)
        dev.adaprox_begin(seqs, scheme=scheme, b2=b2, eps=eps, p=p, check_convergence=check_convergence,
                          prox_max_iter=prox_max_iter, e_rel=e)
        drv = ShardedAdaproxDriver(eng, group, check_convergence, seqs[0].n > 0 or seqs[1].n > 0, prox_max_iter,
                                   dist_module=_collectives(comm, dev, rank, world, group))
        its = drv.run(max_iter, b1)
        dA, dS = dev.get_factors()
        A_local[...] = dA
        S[...] = dS
        r = _lib.Result()
        _lib.check(dev.lib.pmx_iter_result(dev.h, C.byref(r)))
    conv = (bool(r.converged[0]), bool(r.converged[1])) if check_convergence else (None, None)
    return conv, its


def nmf_pgm_sharded(Y_local, A_local, S, M_global, prox_A=None, prox_S=None, accelerated=False, step_scale=1.0,
                    fixed_steps=None, e_rel=1e-3, max_iter=1000, group=None, device=None, comm=None, s_split="auto"):
    """Row-sharded `nmf(Y, A, S, algorithm=pgm, ...)` for one rank (Lipschitz steps x step_scale, or fixed steps).
    [r4] s_split ("auto": whenever N divides by the rank count): the S update sharded too -- reduce-scatter of gS (with A's
    partial Gram matrix and the stopping sums riding in every chunk), each rank updates its N / world columns (and, under
    FISTA, extrapolates them), all-gather of the next evaluation point; any device prox_S (pgm applies it once, row by row of
    S^T).  Returns (converged, iterations)."""
    torch = _lib.require_torch()
    _lib.require_torch()
    import torch.distributed as dist
    from . import operators
    from .engine import DeviceNMF
    prox_A = operators.prox_plus if prox_A is None else prox_A
    prox_S = operators.prox_plus if prox_S is None else prox_S
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = torch.cuda.current_device() if device is None else device
    seqs = [operators.device_proxseq(prox_A, 0), operators.device_proxseq(prox_S, 1)]
    e = (e_rel, e_rel) if np.isscalar(e_rel) else tuple(e_rel)
    tstream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(tstream), DeviceNMF(A_local.shape[0], S.shape[1], A_local.shape[1], device=device,
                                               stream=tstream.cuda_stream) as dev:
        dev.set_Y(Y_local)
        dev.set_factors(A_local, S)
        can_split = world > 1 and S.shape[1] % world == 0
        if s_split is True and not can_split:
            raise NotImplementedError("S-split needs N divisible by the number of ranks")
        eng = ShardEngine(dev, world, rank, M_global, "pgm", s_split=bool(s_split) and can_split)
        dev.pgm_begin(seqs, accelerated=accelerated, step_scale=step_scale, fixed_steps=fixed_steps, e_rel=e)
        eng.bind_eval_buffer()
        loop = ShardedLoop(eng, group, deferred_test=True, dist_module=_collectives(comm, dev, rank, world, group))
        its = loop.run(max_iter)
        dA, dS = dev.get_factors()
        A_local[...] = dA
        S[...] = dS
        r = _lib.Result()
        _lib.check(dev.lib.pmx_iter_result(dev.h, C.byref(r)))
    return (bool(r.converged[0]), bool(r.converged[1])), its


def nmf_bsdmm_sharded(Y_local, A_local, S, M_global, prox_A=None, prox_S=None, proxs_g=None, e_rel=1e-3, e_abs=0.0,
                      max_iter=1000, group=None, device=None, comm=None):
    """Row-sharded `nmf(Y, A, S, algorithm=bsdmm, proxs_g=...)` for one rank.  Returns (converged, iterations)."""
    torch = _lib.require_torch()
    _lib.require_torch()
    import torch.distributed as dist
    from . import operators
    from .engine import DeviceNMF
    prox_A = operators.prox_plus if prox_A is None else prox_A
    prox_S = operators.prox_plus if prox_S is None else prox_S
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = torch.cuda.current_device() if device is None else device
    seq_f = [operators.device_proxseq(prox_A, 0), operators.device_proxseq(prox_S, 1)]
    proxs_g = proxs_g or [None, None]
    seq_g = [None if g is None else [operators.device_proxseq(q, j) for q in g] for j, g in enumerate(proxs_g)]
    er = (e_rel, e_rel) if np.isscalar(e_rel) else tuple(e_rel)
    ea = (e_abs, e_abs) if np.isscalar(e_abs) else tuple(e_abs)
    tstream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(tstream), DeviceNMF(A_local.shape[0], S.shape[1], A_local.shape[1], device=device,
                                               stream=tstream.cuda_stream) as dev:
        dev.set_Y(Y_local)
        dev.set_factors(A_local, S)
        eng = ShardEngine(dev, world, rank, M_global, "bsdmm")
        dev.bsdmm_begin(seq_f, seq_g, e_rel=er, e_abs=ea)
        loop = ShardedLoop(eng, group, deferred_test=False, dist_module=_collectives(comm, dev, rank, world, group))
        its = loop.run(max_iter)
        dA, dS = dev.get_factors()
        A_local[...] = dA
        S[...] = dS
        r = _lib.Result()
        _lib.check(dev.lib.pmx_iter_result(dev.h, C.byref(r)))
    return [bool(r.converged[0]), bool(r.converged[1])], its
