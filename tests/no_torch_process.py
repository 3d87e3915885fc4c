# (run by tests/test_gpu_no_torch.py in a fresh interpreter) the C ABI's collectives from a process that never imports torch ("a row-sharded run without torch")
import ctypes as C, os, sys
os.environ["PMX_TORCH_PRELOAD"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from proxmin_amd import _lib
from proxmin_amd.engine import DeviceNMF
assert "torch" not in sys.modules
with DeviceNMF(512, 768, 16) as dev:
    rng = np.random.default_rng(0)
    A = rng.random((512, 16), dtype=np.float32); S = rng.random((16, 768), dtype=np.float32)
    dev.set_Y((A @ S).astype(np.float32)); dev.set_factors(A, S)
    uid = C.create_string_buffer(128)
    _lib.check(dev.lib.pmx_comm_unique_id(uid))
    _lib.check(dev.lib.pmx_comm_init(dev.h, uid.raw, 0, 1))
    ptr, n = C.c_void_p(), C.c_int64()
    _lib.check(dev.lib.pmx_buffer_ptr(dev.h, _lib.BUF_ST, C.byref(ptr), C.byref(n)))
    _lib.check(dev.lib.pmx_comm_all_reduce(dev.h, ptr, n))            # sum over one rank: S stays S
    dev.sync()
    A2, S2 = dev.get_factors()
    assert np.array_equal(S2, S) and np.array_equal(A2, A)
    assert "torch" not in sys.modules
    print("comm without torch ok: all_reduce over", n.value, "floats on the context's stream; RCCL from", os.environ.get("PMX_RCCL_LIB", "the default search"))
