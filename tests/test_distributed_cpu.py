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
from proxmin_amd.distributed import CommLayout, ShardedAdaproxDriver, shard_rows, HALT_CONVERGED


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
        M, N, K, unity, scheme, pS, check, e_rel, its = case
        Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float64, unity_S=unity, seed=3)
        r0, r1 = shard_rows(M, world)[rank]
        A_l, S = A0[r0:r1].copy(), S0.copy()
        eng = NumpyShardEngine(Y[r0:r1], A_l, S, M, ("plus",), pS, scheme, check, e_rel)
        drv = ShardedAdaproxDriver(eng, None, check, True, 1000, chunk=3)
        n = drv.run(its, np.full(its, 0.9))
        ret[rank] = (r0, r1, A_l, S, n, drv.stopped)
    finally:
        dist.destroy_process_group()


CASES = [
    (61, 90, 4, True, "amsgrad", ("unity_plus", 0), False, 1e-3, 7),
    (50, 64, 3, False, "adam", ("plus",), True, 1e-9, 6),
    (48, 70, 3, False, "amsgrad", ("plus",), True, 8e-2, 60),      # converges early: deferred test must stop at the same iterate
]


@pytest.mark.parametrize("case", CASES)
def test_sharded_protocol_matches_unsharded_oracle(case):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
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
