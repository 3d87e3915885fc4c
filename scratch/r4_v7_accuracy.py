"""gradient error of the split-bf16 variants and the fp16 kernel against fp64 on one problem (relative to max |g| and in Frobenius norm)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd.engine import DeviceNMF
from oracle import nmf_oracle as orc
for (M, N, K, unity, seed) in ((128, 1024, 64, True, 11), (1024, 1536, 64, False, 21), (1024, 1536, 64, True, 21)):
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, unity_S=unity, seed=seed)
    rA, rS = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))
    for mode, var in (("f32", None), ("f16x2", None), ("bf16x3", "7"), ("bf16x3", "5"), ("bf16x3", "4")):
        if var: os.environ["PMX_K1_VARIANT"] = var
        else: os.environ.pop("PMX_K1_VARIANT", None)
        with DeviceNMF(M, N, K, mode=mode) as dev:
            dev.set_Y(Y); dev.set_factors(A, S)
            gA, gS = dev.grad()
            k = dev.k1_info()["kernel"]
        print("%5dx%5dx%d unity=%d %-7s v%-4s %-14s gA max %.2e fro %.2e | gS max %.2e fro %.2e" % (M, N, K, unity, mode, var, k,
              np.abs(gA - rA).max() / np.abs(rA).max(), np.linalg.norm(gA - rA) / np.linalg.norm(rA),
              np.abs(gS - rS).max() / np.abs(rS).max(), np.linalg.norm(gS - rS) / np.linalg.norm(rS)), flush=True)
