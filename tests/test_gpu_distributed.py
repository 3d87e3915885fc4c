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


def test_native_rccl_collectives_through_the_c_abi(pg):
    """include/pmx.h: pmx_comm_* -- RCCL behind the C ABI for a caller without torch.distributed (world 1 on this box: every
    entry point runs, on the context's own stream; sums over one rank are the identity)."""
    import ctypes as C
    import torch
    from proxmin_amd import _lib
    from proxmin_amd.engine import DeviceNMF
    with DeviceNMF(256, 512, 16) as dev:
        uid = C.create_string_buffer(128)
        _lib.check(dev.lib.pmx_comm_unique_id(uid))
        assert any(uid.raw)
        assert dev.lib.pmx_comm_all_reduce(dev.h, C.c_void_p(1), 4) != 0           # before pmx_comm_init: refused, with a message
        assert b"pmx_comm_init" in dev.lib.pmx_last_error()
        _lib.check(dev.lib.pmx_comm_init(dev.h, uid.raw, 0, 1))
        assert dev.lib.pmx_comm_init(dev.h, uid.raw, 0, 1) != 0                    # one communicator per context
        x = torch.arange(4096, dtype=torch.float32, device="cuda") * 0.5 - 7.0
        want = x.clone()
        out = torch.empty(4096, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        _lib.check(dev.lib.pmx_comm_all_reduce(dev.h, C.c_void_p(x.data_ptr()), x.numel()))
        _lib.check(dev.lib.pmx_comm_reduce_scatter(dev.h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), out.numel()))
        g = torch.zeros(4096, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()       # (torch filled these on ITS stream; the collectives run on the context's)
        _lib.check(dev.lib.pmx_comm_all_gather(dev.h, C.c_void_p(out.data_ptr()), C.c_void_p(g.data_ptr()), out.numel()))
        dev.sync()
        assert torch.equal(x, want) and torch.equal(out, want) and torch.equal(g, want)
        _lib.check(dev.lib.pmx_comm_destroy(dev.h))
        _lib.check(dev.lib.pmx_comm_destroy(dev.h))                                # idempotent


@pytest.mark.parametrize("backend", ["adaprox_unity", "adaprox_split_shape", "pgm", "bsdmm"])
def test_world1_native_collectives_equal_torch_collectives(pg, backend):
    """The sharded drivers with comm="native" (NativeRccl: pmx_comm_* on the context's stream) against comm="torch"
    (torch.distributed on the current stream): same kernels, same order -> the same bits."""
    import proxmin_amd as pm
    from proxmin_amd import distributed as pdist
    from oracle import nmf_oracle as orc
    M, N, K = 640, 768, 24
    unity = backend == "adaprox_unity"
    Y, A0, S0 = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=11)
    res = {}
    for comm in ("torch", "native"):
        A, S = A0.copy(), S0.copy()
        if backend.startswith("adaprox"):
            pS = partial(pm.operators.prox_unity_plus, axis=0) if unity else pm.operators.prox_plus
            r = pdist.nmf_adaprox_sharded(Y, A, S, M, prox_A=pm.operators.prox_plus, prox_S=pS, scheme="amsgrad",
                                          check_convergence=False, e_rel=1e-3, max_iter=6, comm=comm)
        elif backend == "pgm":
            r = pdist.nmf_pgm_sharded(Y, A, S, M, accelerated=True, e_rel=1e-9, max_iter=6, comm=comm)
        else:
            pg_ = [[pm.operators.prox_plus, partial(pm.operators.prox_soft, thresh=0.01)]] * 2
            r = pdist.nmf_bsdmm_sharded(Y, A, S, M, proxs_g=pg_, e_rel=1e-9, max_iter=5, comm=comm)
        res[comm] = (A, S, r[1])
    assert res["torch"][2] == res["native"][2]
    assert np.array_equal(res["torch"][0], res["native"][0]) and np.array_equal(res["torch"][1], res["native"][1])
