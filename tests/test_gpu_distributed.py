"""GPU, world_size 1 over RCCL ("nccl" backend): the row-sharded driver + HIP phase entry points must
reproduce the single-GPU nmf() result (same kernels; gS takes the detour through the comm buffer and
a real all-reduce).  Multi-rank behaviour of the protocol itself is covered on CPU with gloo
(tests/test_distributed_cpu.py); 8-GPU runs are the driver's."""
import os
import socket
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("unity,scheme,check,e_rel,its", [(True, "amsgrad", False, 1e-3, 7), (False, "adam", True, 1e-9, 6),
                                                          (False, "amsgrad", True, 5e-2, 80)])
def test_world1_sharded_equals_single_gpu(pg, unity, scheme, check, e_rel, its):
    import __graft_entry__ as g
    g.build()
    import proxmin_amd as pm
    from proxmin_amd import distributed as pdist
    from oracle import nmf_oracle as orc
    M, N, K = 700, 900, 24
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=4)
    pS = partial(pm.operators.prox_unity_plus, axis=0) if unity else pm.operators.prox_plus
    A1, S1 = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    ret = pm.nmf.nmf(Y, A1, S1, algorithm=pm.adaprox, scheme=scheme, prox_S=pS, max_iter=its, e_rel=e_rel,
                     check_convergence=check, callback=tb)
    A2, S2 = A0.copy(), S0.copy()
    conv, n = pdist.nmf_adaprox_sharded(Y, A2, S2, M, prox_A=pm.operators.prox_plus, prox_S=pS, scheme=scheme,
                                        check_convergence=check, e_rel=e_rel, max_iter=its)
    assert n == len(tb.trace)
    np.testing.assert_allclose(A2, A1, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(S2, S1, rtol=2e-5, atol=2e-6)
    if check:
        assert conv == ret[0]


@pytest.mark.parametrize("accel,e_rel,its", [(False, 1e-9, 7), (True, 1e-9, 6), (False, 3e-2, 300)])
def test_world1_sharded_pgm_equals_single_gpu(pg, accel, e_rel, its):
    import proxmin_amd as pm
    from proxmin_amd import distributed as pdist
    from oracle import nmf_oracle as orc
    M, N, K = 520, 700, 12
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=6)
    scale = 0.5 if accel else 1.0
    A1, S1 = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv1, _, _ = pm.nmf.nmf(Y, A1, S1, step=pm.nmf.scaled_step_pgm(scale), accelerated=accel, max_iter=its, e_rel=e_rel, callback=tb)
    A2, S2 = A0.copy(), S0.copy()
    conv2, n = pdist.nmf_pgm_sharded(Y, A2, S2, M, accelerated=accel, step_scale=scale, e_rel=e_rel, max_iter=its)
    assert n == len(tb.trace) and conv2 == conv1
    np.testing.assert_allclose(A2, A1, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(S2, S1, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("e_rel,its", [(1e-9, 6), (0.5, 60)])
def test_world1_sharded_bsdmm_equals_single_gpu(pg, e_rel, its):
    import proxmin_amd as pm
    from proxmin_amd import distributed as pdist
    from oracle import nmf_oracle as orc
    ops = pm.operators
    M, N, K = 480, 640, 10
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, seed=8)
    pgl = [[ops.prox_plus, partial(ops.prox_soft, thresh=0.01)], [ops.prox_plus, partial(ops.prox_soft, thresh=0.01)]]
    A1, S1 = A0.copy(), S0.copy()
    tb = pm.utils.Traceback()
    conv1 = pm.nmf.nmf(Y, A1, S1, algorithm=pm.bsdmm, proxs_g=pgl, max_iter=its, e_rel=e_rel, callback=tb)
    A2, S2 = A0.copy(), S0.copy()
    conv2, n = pdist.nmf_bsdmm_sharded(Y, A2, S2, M, proxs_g=pgl, e_rel=e_rel, max_iter=its)
    assert n == len(tb.trace) and conv2 == conv1
    np.testing.assert_allclose(A2, A1, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(S2, S1, rtol=2e-5, atol=2e-6)
