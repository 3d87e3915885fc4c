# scratch: randomized parity sweep of K1 (all three arithmetic modes) against the fp64 oracle
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
from proxmin_amd.engine import DeviceNMF
from oracle import nmf_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(n_cases):
    kind = rng.integers(0, 8)
    if kind == 4:      # K = 128 whole blocks (k_grad_f16_k128 in mode f16x2)
        M, N, K = 128 * int(rng.integers(1, 40)), 128 * int(rng.integers(1, 24)), 128
    elif kind >= 6:    # [r4] ragged shapes with a tuned K: the zero-padded frame (pmx_k1_frame)
        M, N, K = int(rng.integers(300, 7000)), int(rng.integers(300, 7000)), int(rng.choice([32, 64, 128]))
    elif kind == 5:    # small problems (k_grad_small<8|16>)
        M, N, K = int(rng.integers(1, 1500)), int(rng.integers(1, 700)), int(rng.integers(1, 17))
    elif kind == 0:      # anything
        M, N, K = int(rng.integers(1, 2500)), int(rng.integers(1, 2500)), int(rng.integers(1, 129))
    elif kind == 1:    # K = 64, whole blocks (fast kernels)
        M, N, K = 128 * int(rng.integers(1, 24)), 256 * int(rng.integers(1, 12)), 64
    elif kind == 2:    # K = 64, N multiple of 64 only / ragged M
        M, N, K = int(rng.integers(1, 3000)), 64 * int(rng.integers(1, 40)), 64
    else:              # K = 32 whole blocks
        M, N, K = 128 * int(rng.integers(1, 16)), 64 * int(rng.integers(1, 30)), 32
    Y, A, S = orc.synthetic_problem(M, N, K, np.float32, seed=int(rng.integers(1 << 30)))
    scale = 10.0 ** rng.uniform(-2, 2)
    A = (A * scale).astype(np.float32)
    useW = rng.random() < 0.25
    W = (0.1 + rng.random((M, N))).astype(np.float32) if useW else None
    x64 = [A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64)] + ([W.astype(np.float64)] if useW else [])
    rA, rS = orc.residual_gradients(*x64)
    rl = orc.half_sq_residual(*x64)
    for mode in ("f32", "bf16x3", "f16x2"):
        try:
            with DeviceNMF(M, N, K, mode=mode) as dev:
                dev.set_Y(Y)
                if useW:
                    try:
                        dev.set_W(W)
                    except NotImplementedError:
                        continue
                dev.set_factors(A, S)
                gA, gS = dev.grad()
                loss = dev.loglike()
            eA = np.abs(gA - rA).max() / max(np.abs(rA).max(), 1e-30)
            eS = np.abs(gS - rS).max() / max(np.abs(rS).max(), 1e-30)
            el = abs(loss - rl) / max(rl, 1e-30)
            ok = eA < 2e-5 and eS < 2e-5 and el < 2e-5
        except Exception as e:
            ok = False; eA = eS = el = float("nan"); print("EXC", repr(e))
        if not ok:
            bad += 1
            print("FAIL case %d %dx%dx%d W=%d %s: gA %.2e gS %.2e loss %.2e" % (case, M, N, K, useW, mode, eA, eS, el), flush=True)
print("fuzz done: %d cases x 3 modes, %d failures" % (n_cases, bad), flush=True)
