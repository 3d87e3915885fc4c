"""CPU, world_size 2, gloo: the row-sharded protocol (what is packed into the single all-reduce, the
global step rule for A, the deferred outer stopping test) reproduces the unsharded oracle.

The HIP engine cannot run here (no GPU), so the driver -- the same `ShardedAdaproxDriver` the GPU path
uses -- is exercised with a NumPy stand-in engine built from the oracle's pieces.  On the GPU the same
driver is covered at world_size 1 by tests/test_gpu_distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nmf_oracle as orc
from proxmin_amd.distributed import CommLayout, SplitLayout, ShardedAdaproxDriver, ShardedLoop, shard_rows, HALT_CONVERGED


class NumpyShardEngine:
    """Stand-in for proxmin_amd.distributed.ShardEngine with identical phase semantics (fp64 NumPy)."""

    def __init__(self, Y_l, A_l, S, M_global, prox_A, prox_S, scheme, check, e_rel, b2=0.999, eps=1e-8, p=0.25):
        self.Y, self.A, self.S = Y_l, A_l, S
        self.Mg = M_global
        self.K, self.N = S.shape
        self.lay = CommLayout(self.N, self.K)
        self.comm = torch.zeros(self.lay.count, dtype=torch.float64)
        self.spec = [prox_A, prox_S]
        self.scheme, self.check, self.e = scheme, check, e_rel
        self.b2, self.eps, self.p = b2, eps, p
        self.Mm = [np.zeros_like(A_l), np.zeros_like(S)]
        self.Vv = [np.zeros_like(A_l), np.zeros_like(S)]
        self.halted, self.reason, self.it_done = 0, 0, 0
        self.sums = [(0.0, 0.0), (0.0, 0.0)]
        self.tau = [0, 0]

    def _pack_extras(self):
        L, c = self.lay, self.comm.numpy()
        c[L.gram:L.colsum] = 0
        c[L.colsum:L.colsum + self.K] = self.A.sum(0)
        c[L.scalars:L.scalars + 2] = self.sums[0]

    def _post(self, have_prev):
        L, c = self.lay, self.comm.numpy()
        self.alpha = [c[L.colsum:L.colsum + self.K] / self.Mg / 10, self.S.mean(axis=1)[:, None] / 10]
        if self.check and have_prev:
            dA, nA = c[L.scalars], c[L.scalars + 1]
            dS, nS = self.sums[1]
            if dA <= self.e ** 2 * nA and dS <= self.e ** 2 * nS:
                self.halted, self.reason = 1, HALT_CONVERGED

    def phase(self, phase, it, b1_it, b1_prev, nsub):
        if self.halted:
            return
        L, c = self.lay, self.comm.numpy()
        if phase == 0:
            self.gA, gS = orc.residual_gradients(self.A, self.S, self.Y)
            c[:L.gram] = gS.T.ravel()
            self._pack_extras()
        elif phase == 1:
            self._post(it > 0)
            if self.halted:
                return
            G = [self.gA, c[:L.gram].reshape(self.N, self.K).T.copy()]
            X = [self.A, self.S]
            b1 = np.full(it + 1, b1_it)
            b1[it - 1] = b1_prev
            for j in range(2):
                prev = X[j].copy()
                Phi, Psi = orc.moment_update(self.scheme, it, G[j], self.Mm[j], self.Vv[j], None, b1, self.b2, self.eps, self.p)
                X[j][:] -= self.alpha[j] * Phi / Psi
                if self.spec[j] is not None:
                    z = X[j].copy()
                    gamma = self.alpha[j] / np.max(Psi) if j == 1 else self.alpha[j]   # A: projection, gamma irrelevant
                    for tau in range(1, 1001):
                        zn = orc.apply_prox(z - (gamma / self.alpha[j] * Psi if j == 1 else 1.0) * (z - X[j]), gamma, self.spec[j])
                        done = ((zn - z) ** 2).sum() <= self.e ** 2 * (z ** 2).sum()
                        z = zn
                        if done:
                            break
                    self.tau[j] = tau
                    X[j][:] = z
                self.sums[j] = (float(((X[j] - prev) ** 2).sum()), float((X[j] ** 2).sum()))
            self.it_done += 1
        elif phase == 2:
            self._pack_extras()
        elif phase == 3:
            self._post(True)

    def chain_status(self):
        return self.halted, self.reason, self.it_done, tuple(self.tau)

    def more_subs(self, t0, n):
        raise AssertionError("the stand-in engine never runs out of sub-iteration passes")


class NumpySplitShardEngine(NumpyShardEngine):
    """S-split protocol (pmx_set_s_split): the comm buffer is `world` chunks [gSt rows of rank q | Gram | colsum(A) | colsum(S) |
    scalars]; after the driver's reduce-scatter this rank updates A's local rows and ITS columns of S (moments included), the
    driver all-gathers S^T.  Same phase semantics as the HIP engine's (fp64 NumPy)."""

    s_split = True

    def __init__(self, rank, world, *args, **kw):
        super().__init__(*args, **kw)
        self.rank, self.world = rank, world
        self.lay = SplitLayout(self.N, self.K, world)
        self.comm = torch.zeros(self.lay.count, dtype=torch.float64)
        self.comm_out = torch.zeros(self.lay.chunk, dtype=torch.float64)
        self.c0 = rank * self.lay.sncol
        self.c1 = self.c0 + self.lay.sncol
        self.st_full = torch.from_numpy(np.ascontiguousarray(self.S.T))      # S^T (N x K), what the driver all-gathers
        self.St = self.st_full.numpy()
        self.MmS = np.zeros((self.K, self.lay.sncol))
        self.VvS = np.zeros((self.K, self.lay.sncol))
        self.sumsS = (0.0, 0.0)
        self.first = True

    def _extras_split(self):
        L, c = self.lay, self.comm.numpy()
        Sloc = self.St[self.c0:self.c1].T                          # K x sncol
        for q in range(self.world):
            b = q * L.chunk
            c[b + L.gram:b + L.colsum_A] = 0
            c[b + L.colsum_A:b + L.colsum_A + self.K] = self.A.sum(0)
            c[b + L.colsum_S:b + L.colsum_S + self.K] = Sloc.sum(1)
            c[b + L.scalars:b + L.scalars + 4] = (*self.sums[0], *self.sumsS)

    def _post_split(self, have_prev):
        L, o = self.lay, self.comm_out.numpy()
        self.alpha = [o[L.colsum_A:L.colsum_A + self.K] / self.Mg / 10, (o[L.colsum_S:L.colsum_S + self.K] / self.N / 10)[:, None]]
        if self.check and have_prev:
            dA, nA, dS, nS = o[L.scalars:L.scalars + 4]
            if dA <= self.e ** 2 * nA and dS <= self.e ** 2 * nS:
                self.halted, self.reason = 1, HALT_CONVERGED

    def phase(self, phase, it, b1_it, b1_prev, nsub):
        if self.halted:
            return
        L, c = self.lay, self.comm.numpy()
        if phase == 0:
            S = np.ascontiguousarray(self.St.T)                    # the all-gathered S
            self.gA, gS = orc.residual_gradients(self.A, S, self.Y)
            gSt = gS.T
            for q in range(self.world):
                c[q * L.chunk:q * L.chunk + L.gram] = gSt[q * L.sncol:(q + 1) * L.sncol].ravel()
            self._extras_split()
        elif phase == 1:
            self._post_split(it > 0)
            if self.halted:
                return
            o = self.comm_out.numpy()
            b1 = np.full(it + 1, b1_it)
            b1[it - 1] = b1_prev
            Sloc = np.ascontiguousarray(self.St[self.c0:self.c1].T)
            G = [self.gA, o[:L.gram].reshape(L.sncol, self.K).T.copy()]
            X = [self.A, Sloc]
            Mm, Vv = [self.Mm[0], self.MmS], [self.Vv[0], self.VvS]
            for j in range(2):
                prev = X[j].copy()
                Phi, Psi = orc.moment_update(self.scheme, it, G[j], Mm[j], Vv[j], None, b1, self.b2, self.eps, self.p)
                X[j][:] -= self.alpha[j] * Phi / Psi
                if self.spec[j] is not None:                       # projection-type operators only: one pass is the fixed point
                    X[j][:] = orc.apply_prox(X[j], self.alpha[j], self.spec[j])
                    self.tau[j] = 2
                sums = (float(((X[j] - prev) ** 2).sum()), float((X[j] ** 2).sum()))
                if j == 0:
                    self.sums[0] = sums
                else:
                    self.sumsS = sums
            self.St[self.c0:self.c1] = Sloc.T
            self.it_done += 1
        elif phase == 2:
            self._extras_split()
        elif phase == 3:
            self._post_split(True)


class NumpyPgmShardEngine:
    """pgm phases (FISTA optional), same protocol as pmx_pgm_phase."""

    def __init__(self, Y_l, A_l, S, M_global, accelerated, e_rel):
        self.Y, self.A, self.S, self.Mg = Y_l, A_l, S, M_global
        self.K, self.N = S.shape
        self.lay = CommLayout(self.N, self.K)
        self.comm = torch.zeros(self.lay.count, dtype=torch.float64)
        self.acc, self.e = accelerated, e_rel
        self.omegas = orc.nesterov_omegas(10000, accelerated)
        self.prev = None
        self.halted, self.reason, self.it_done = 0, 0, 0
        self.sums = [(0.0, 0.0), (0.0, 0.0)]

    def _extras(self):
        L, c = self.lay, self.comm.numpy()
        c[L.scalars:L.scalars + 2] = self.sums[0]

    def _test(self):
        L, c = self.lay, self.comm.numpy()
        dA, nA = c[L.scalars], c[L.scalars + 1]
        dS, nS = self.sums[1]
        if dA <= self.e ** 2 * nA and dS <= self.e ** 2 * nS:
            self.halted, self.reason = 1, HALT_CONVERGED

    def phase(self, phase, it, *unused):
        if self.halted:
            return
        L, c = self.lay, self.comm.numpy()
        K, KP = self.K, self.lay.KP
        if phase == 0:
            om = self.omegas[it]
            X = [self.A, self.S]
            self.E = [X[j] + om * (X[j] - self.prev[j]) for j in range(2)] if om > 0 else [x.copy() for x in X]
            self.gA, gS = orc.residual_gradients(self.E[0], self.E[1], self.Y)
            c[:L.gram] = gS.T.ravel()
            G = np.zeros((KP, KP))
            G[:K, :K] = self.E[0].T @ self.E[0]
            c[L.gram:L.colsum] = G.ravel()
            self._extras()
        elif phase == 1:
            if it > 0:
                self._test()
                if self.halted:
                    return
            G = c[L.gram:L.colsum].reshape(KP, KP)[:K, :K]
            sS = 1 / np.linalg.eigvalsh(G)[-1]
            sA = 1 / orc.gram_lambda_max(self.E[1].T)
            X = [self.A, self.S]
            self.prev = [x.copy() for x in X]
            Gs = [self.gA, c[:L.gram].reshape(self.N, K).T.copy()]
            for j, st in enumerate((sA, sS)):
                X[j][:] = orc.apply_prox(self.E[j] - st * Gs[j], st, ("plus",))
                self.sums[j] = (float(((X[j] - self.prev[j]) ** 2).sum()), float((X[j] ** 2).sum()))
            self.it_done += 1
        elif phase == 2:
            self._extras()
        elif phase == 3:
            self._test()

    def chain_status(self):
        return self.halted, self.reason, self.it_done, (0, 0)


class NumpyBsdmmShardEngine:
    """bsdmm phases with proxs_g = [plus, soft(thresh)] on both blocks, same protocol as pmx_bsdmm_phase."""

    def __init__(self, Y_l, A_l, S, M_global, thresh, e_rel):
        self.Y, self.A, self.S, self.Mg = Y_l, A_l, S, M_global
        self.K, self.N = S.shape
        self.lay = CommLayout(self.N, self.K)
        self.comm = torch.zeros(self.lay.count, dtype=torch.float64)
        self.pg = [("plus",), ("soft", thresh, "relative")]
        self.e = e_rel
        self.Z = [[A_l.copy(), A_l.copy()], [S.copy(), S.copy()]]
        self.U = [[np.zeros_like(A_l), np.zeros_like(A_l)], [np.zeros_like(S), np.zeros_like(S)]]
        self.halted, self.reason, self.it_done = 0, 0, 0

    def _block(self, j, sf, G):
        X, Z, U = (self.A, self.S)[j], self.Z[j], self.U[j]
        sg = sf * 2 * 2
        dX = sum(sf / sg * (X - Z[i] + U[i]) for i in range(2))
        old = X.copy()
        X[:] = orc.apply_prox((X - dX) - sf * G, sf, ("plus",))
        sums = [float(((X - old) ** 2).sum()), float((X ** 2).sum())]
        for i in range(2):
            Zn = orc.apply_prox(X + U[i], sg, self.pg[i])
            R = X - Zn
            Sd = -1 / sg * (Zn - Z[i])
            Z[i][:] = Zn
            U[i][:] += R
            sums += [float((R ** 2).sum()), float((Sd ** 2).sum()), float((Zn ** 2).sum()), float(((U[i] / sg) ** 2).sum())]
        return sums

    def _conv(self, sums, size):
        x2 = sums[1]
        ok = True
        for i in range(2):
            r2, s2, z2, u2 = sums[2 + 4 * i: 6 + 4 * i]
            e_pri = self.e * max(np.sqrt(x2), np.sqrt(z2))
            e_dual = self.e * np.sqrt(u2)
            ok &= (np.sqrt(r2) <= e_pri) and (np.sqrt(s2) <= e_dual)
        return bool(ok)

    def phase(self, phase, it, *unused):
        if self.halted:
            return
        L, c = self.lay, self.comm.numpy()
        K, KP = self.K, self.lay.KP
        if phase == 0:
            sA = 1 / orc.gram_lambda_max(self.S.T)
            gA, _ = orc.residual_gradients(self.A, self.S, self.Y)
            sums = self._block(0, sA, gA)
            _, gS = orc.residual_gradients(self.A, self.S, self.Y)
            c[:L.gram] = gS.T.ravel()
            G = np.zeros((KP, KP))
            G[:K, :K] = self.A.T @ self.A
            c[L.gram:L.colsum] = G.ravel()
            c[L.scalars:L.scalars + len(sums)] = sums
        else:
            G = c[L.gram:L.colsum].reshape(KP, KP)[:K, :K]
            sS = 1 / np.linalg.eigvalsh(G)[-1]
            convA = self._conv(list(c[L.scalars:L.scalars + 10]), self.Mg * K)
            sumsS = self._block(1, sS, c[:L.gram].reshape(self.N, K).T.copy())
            convS = self._conv(sumsS, self.N * K)
            self.it_done += 1
            if convA and convS:
                self.halted, self.reason = 1, HALT_CONVERGED

    def chain_status(self):
        return self.halted, self.reason, self.it_done, (0, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case[0] in ("pgm", "bsdmm"):
            alg, M, N, K, opt, e_rel, its = case
            Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, seed=3)
            r0, r1 = shard_rows(M, world)[rank]
            A_l, S = A0[r0:r1].copy(), S0.copy()
            eng = (NumpyPgmShardEngine(Y[r0:r1], A_l, S, M, opt, e_rel) if alg == "pgm"
                   else NumpyBsdmmShardEngine(Y[r0:r1], A_l, S, M, opt, e_rel))
            loop = ShardedLoop(eng, None, deferred_test=(alg == "pgm"), chunk=3)
            n = loop.run(its)
            ret[rank] = (r0, r1, A_l, S, n, loop.stopped)
            return
        split = case[0] == "split"
        if split:
            case = case[1:]
        M, N, K, unity, scheme, pS, check, e_rel, its = case
        Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=unity, seed=3)
        r0, r1 = shard_rows(M, world)[rank]
        A_l, S = A0[r0:r1].copy(), S0.copy()
        if split:
            eng = NumpySplitShardEngine(rank, world, Y[r0:r1], A_l, S, M, ("plus",), pS, scheme, check, e_rel)
        else:
            eng = NumpyShardEngine(Y[r0:r1], A_l, S, M, ("plus",), pS, scheme, check, e_rel)
        drv = ShardedAdaproxDriver(eng, None, check, True, 1000, chunk=3)
        n = drv.run(its, np.full(its, 0.9))
        if split:
            S = np.ascontiguousarray(eng.St.T)
        ret[rank] = (r0, r1, A_l, S, n, drv.stopped)
    finally:
        dist.destroy_process_group()


CASES = [
    (61, 90, 4, True, "amsgrad", ("unity_plus", 0), False, 1e-3, 7),
    (50, 64, 3, False, "adam", ("plus",), True, 1e-9, 6),
    (48, 70, 3, False, "amsgrad", ("plus",), True, 8e-2, 60),      # converges early: deferred test must stop at the same iterate
    # S-split: reduce-scatter -> each rank updates its N / 2 columns of S -> all-gather (driver's gloo fall-backs included)
    ("split", 50, 64, 3, False, "adam", ("plus",), True, 1e-9, 6),
    ("split", 48, 70, 3, False, "amsgrad", ("plus",), True, 8e-2, 60),
    ("split", 61, 90, 4, False, "amsgrad", ("plus",), False, 1e-3, 7),
]


@pytest.mark.parametrize("case", CASES)
def test_sharded_protocol_matches_unsharded_oracle(case):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    if case[0] == "split":
        case = case[1:]
    M, N, K, unity, scheme, pS, check, e_rel, its = case
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=unity, seed=3)
    Ao, So = A0.copy(), S0.copy()
    conv, _, _, _, n_ref, _ = orc.adaprox_nmf(Y, Ao, So, ("plus",), pS, scheme=scheme, max_iter=its, e_rel=e_rel, check_convergence=check)
    A = np.zeros_like(A0)
    for rank in range(world):
        r0, r1, A_l, S, n, stopped = ret[rank]
        A[r0:r1] = A_l
        np.testing.assert_allclose(S, So, rtol=1e-9, atol=1e-12)          # replicated S identical to the unsharded run
        assert n == n_ref
        if check:
            assert stopped == all(conv)
    np.testing.assert_allclose(A, Ao, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(ret[0][3], ret[1][3])                   # bitwise identical replicas


def test_shard_rows_and_layout():
    assert shard_rows(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard_rows(16384, 8)[-1] == (14336, 16384)
    L = CommLayout(1000, 5)
    assert (L.gram, L.colsum, L.scalars, L.count) == (5000, 5000 + 32 * 32, 5000 + 1024 + 128, 5000 + 1024 + 128 + 32)
    P = SplitLayout(1000, 5, 4)
    assert (P.sncol, P.gram, P.colsum_A, P.colsum_S, P.scalars, P.chunk, P.count) == (250, 1250, 1250 + 1024, 1250 + 1152, 1250 + 1280, 1250 + 1312, 4 * (1250 + 1312))


PB_CASES = [
    ("pgm", 57, 80, 4, False, 1e-9, 7),
    ("pgm", 48, 66, 3, True, 1e-9, 6),          # FISTA (undamped steps; first iterations only)
    ("pgm", 40, 60, 3, False, 6e-2, 200),       # converges early: deferred test stops at the same iterate
    ("bsdmm", 52, 70, 4, 0.01, 1e-9, 6),
    ("bsdmm", 44, 50, 3, 0.01, 0.5, 50),        # converges early (loose e_rel)
]


@pytest.mark.parametrize("case", PB_CASES)
def test_sharded_pgm_bsdmm_match_unsharded_oracle(case):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    alg, M, N, K, opt, e_rel, its = case
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, seed=3)
    Ao, So = A0.copy(), S0.copy()
    if alg == "pgm":
        conv, _, _, n_ref = orc.pgm_nmf(Y, Ao, So, accelerated=opt, max_iter=its, e_rel=e_rel)
    else:
        pg = [[("plus",), ("soft", opt, "relative")]] * 2
        conv, n_ref = orc.bsdmm_nmf(Y, Ao, So, proxs_g=pg, max_iter=its, e_rel=e_rel)
    A = np.zeros_like(A0)
    for rank in range(world):
        r0, r1, A_l, S, n, stopped = ret[rank]
        A[r0:r1] = A_l
        np.testing.assert_allclose(S, So, rtol=1e-8, atol=1e-11)
        assert n == n_ref, (n, n_ref)
        assert stopped == all(conv)
    np.testing.assert_allclose(A, Ao, rtol=1e-8, atol=1e-11)


def test_native_bootstrap_failure_reaches_every_rank():
    """NativeRccl.bootstrap when rank 0 cannot draw a communicator id (RCCL not loadable).  Rank 0 must STILL take part in the
    broadcast (the other ranks are waiting in it) and hand them a marker; every rank then raises without entering pmx_comm_init, so the
    caller's agreement all-reduce (distributed._collectives) is reached by all of them instead of deadlocking."""
    from proxmin_amd import _lib
    from proxmin_amd.distributed import NativeRccl

    class FakeLib:
        def __init__(self, fail):
            self.fail, self.init_calls = fail, 0

        def pmx_comm_unique_id(self, buf):
            return 3 if self.fail else 0

        def pmx_comm_init(self, *a):
            self.init_calls += 1
            return 0

    class FakeDev:
        def __init__(self, fail):
            self.lib, self.h = FakeLib(fail), None

    sent = []

    def bcast_rank0(raw):
        sent.append(raw)
        return raw

    real_check = _lib.check
    try:
        def check(rc):                       # (no libpmx.so needed on this path: the fake reports its own failure)
            if rc != 0:
                raise _lib.PmxError("libpmx error %d: rccl not loadable" % rc)
        _lib.check = check
        dev0 = FakeDev(True)
        with pytest.raises(_lib.PmxError):
            NativeRccl.bootstrap(dev0, 0, 2, bcast_rank0)
        assert sent == [b""] and dev0.lib.init_calls == 0          # the broadcast happened, with the marker; no communicator was entered
        dev1 = FakeDev(False)
        with pytest.raises(_lib.PmxError):
            NativeRccl.bootstrap(dev1, 1, 2, lambda raw: sent[0])    # rank 1 receives the marker
        assert dev1.lib.init_calls == 0
        ok0 = FakeDev(False)
        nat = NativeRccl.bootstrap(ok0, 0, 1, lambda raw: raw)       # the good path still joins
        assert isinstance(nat, NativeRccl) and ok0.lib.init_calls == 1
    finally:
        _lib.check = real_check
