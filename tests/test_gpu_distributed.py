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
