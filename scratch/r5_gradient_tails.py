# scratch: WHERE mode f16x2r's extra out-of-tolerance entries (1.8 x the NumPy-fp32 yardstick in S at full cfg3, exact fp32 on the device 1.1 x) come from.
# Full cfg3, state after 2 adaprox iterations (mode f32): the gradients of that state in every mode against fp64 on 512 rows of A / 512 columns
# of S -- rms, bias, and the TAIL of the absolute error (the eps-clamp entries amplify absolute gradient error by up to 1e4).  NumPy fp32 beside them.
import sys, os
os.environ.setdefault("PMX_TORCH_PRELOAD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd import engine as eng
M, N, K, backend, unity, _ = bench.CONFIGS["cfg3"]
Yd, A0, S0 = bench.make_problem_device(M, N, K, unity, 4321, torch.device("cuda", 0))
with eng.DeviceNMF(M, N, K, mode="f32") as dev:
    dev.set_Y_device(Yd.data_ptr(), ld=N, copy=False, keepalive=Yd); dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, backend, unity); run(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
    A, S = dev.get_factors()
rng = np.random.default_rng(5)
rows = np.sort(rng.choice(M, 512, replace=False)); cols = np.sort(rng.choice(N, 512, replace=False))
A64, S64 = A.astype(np.float64), S.astype(np.float64)
Yr = Yd[torch.as_tensor(rows, device=Yd.device)].cpu().numpy(); Yc = Yd[:, torch.as_tensor(cols, device=Yd.device)].cpu().numpy()
rA = (A64[rows] @ S64 - Yr) @ S64.T; rS = A64.T @ (A64 @ S64[:, cols] - Yc)
def stats(name, gA, gS):
    out = []
    for gq, r in ((gA, rA), (gS, rS)):
        e = gq.astype(np.float64) - r
        sc = np.abs(r).max()
        out.append("rms %.2e bias %+.1e q99.9 %.2e max %.2e" % (np.sqrt((e ** 2).mean()) / sc, e.mean() / sc, np.quantile(np.abs(e), 0.999) / sc, np.abs(e).max() / sc))
    print("%-22s gA[rows]: %s | gS[:, cols]: %s" % (name, out[0], out[1]), flush=True)
stats("numpy fp32", (A[rows] @ S - Yr) @ S.T, A.T @ (A @ S[:, cols] - Yc))
for name, mode, env in (("f32", "f32", {}), ("f16x2", "f16x2", {}), ("f16x2r <R3>", "f16x2r", {"PMX_F16_R3": "1"}), ("f16x2r <HH>", "f16x2r", {"PMX_F16_R3": "2"})):
    os.environ.update(env)
    with eng.DeviceNMF(M, N, K, mode=mode) as dev:
        dev.set_Y_device(Yd.data_ptr(), ld=N, copy=False, keepalive=Yd); dev.set_factors(A, S)
        gA, gS = dev.grad()
    for k in env: del os.environ[k]
    stats(name + " (device)", gA[rows], gS[:, cols])
    # is the bias a UNIFORM offset of the residual?  gS error ~ -beta colsum(A)[k], gA error ~ -beta rowsum(S)[k] with ONE beta
    eS = (gS[:, cols].astype(np.float64) - rS).mean(axis=1); eA = (gA[rows].astype(np.float64) - rA).mean(axis=0)
    bS = eS / A64.sum(axis=0); bA = eA / S64.sum(axis=1)
    print("    implied residual offset: from gS %.3e +- %.1e, from gA %.3e +- %.1e  (mean |P| %.3g, ulp %.2e)" % (bS.mean(), bS.std(), bA.mean(), bA.std(), float((A64[rows] @ S64).mean()), 2.0 ** (np.floor(np.log2((A64[rows] @ S64).mean())) - 23)), flush=True)
