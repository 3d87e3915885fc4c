# scratch (CPU, NumPy): what each split arithmetic costs in accuracy at full cfg3, emulated -- the oracle's adaprox / AMSGrad + prox_unity_plus for 3 iterations
# with its gradient function replaced by an emulation of the device's products (fp16 / bf16 terms held in fp32 arrays, products exact, fp32 accumulate),
# against the fp64 oracle; the oracle's own fp32 run is the yardstick (tests/test_gpu_parity_long.py measures the same on the device).
#   f32      numpy fp32 GEMMs (the yardstick)
#   f16x2    what k_grad_f16_v8 computes: two fp16 terms per operand, 3 products per contraction (low x low dropped), power-of-two scales
#   f16x2+ll the same with the low x low product in A S
#   f16r3    THREE fp16 terms for A and S in A S (6 products), two in the gradient contractions: the residual to ~33 bits
#   f16r3g3  ... and three terms of R in the gradient contractions as well
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import nmf_oracle as orc

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K, ITS = 64, 3
f32 = np.float32


def terms16(x, n):
    out, r = [], x
    for _ in range(n):
        t = r.astype(np.float16).astype(f32)
        out.append(t)
        r = r - t
    return out


def scale_of(m, top=14):
    q = np.frexp(f32(m))[1]
    return f32(np.ldexp(1.0, int(top - q))) if m > 0 else f32(1)       # (np.float32 ** np.int64 would promote to float64)


def make_grad(nP, drop_ll, nR, nG=2, gmax=None, balance=False, split_acc=0, gram_fix=0):
    def grad(A, S, Y, W=None):
        A, S = A.astype(f32), S.astype(f32)
        if gram_fix:
            # the two-term gradients corrected to first order with K x K Gram matrices instead of wider products: the residual was computed
            # from A - dA and S - dS (dA, dS: what two fp16 terms leave of the operands), so  gA += dA (S S^T),  gS += (A^T A) dS
            # (gram_fix = 2: the cross terms A (dS S^T) and (A^T dA) S as well)
            gA, gS = make_grad(2, True, 2)(A, S, Y)
            sA, sS = scale_of(np.abs(A).max()), scale_of(np.abs(S).max())
            dA = (A * sA - sum(terms16(A * sA, 2))) / sA
            dS = (S * sS - sum(terms16(S * sS, 2))) / sS
            gA = gA + dA @ (S @ S.T)
            gS = gS + (A.T @ A) @ dS
            if gram_fix >= 2:
                gA = gA + A @ (dS @ S.T)
                gS = gS + (A.T @ dA) @ S
            return gA.astype(f32), gS.astype(f32)
        if split_acc:
            # TWO accumulators in A S: the high x high product in one, the small products (h l + l h [+ h 3 + 3 h: third terms of S and
            # of A, split_acc = 2] [+ l l, split_acc = 3]) in another, R = (P_hi - Y) + P_lo: what is below half an ulp of P survives
            sA, sS = scale_of(np.abs(A).max()), scale_of(np.abs(S).max())
            a, s = terms16(A * sA, 3), terms16(S * sS, 3)
            u = f32(1) / (sA * sS)
            lo = a[0] @ s[1] + a[1] @ s[0]
            if split_acc >= 2:
                lo += a[0] @ s[2] + a[2] @ s[0]
            if split_acc >= 3:
                lo += a[1] @ s[1]
            R = ((a[0] @ s[0]) * u - Y) + lo * u
            del lo
            sR = scale_of(np.abs(Y).max() + K * np.abs(A).max() * np.abs(S).max())
            r = terms16(R * sR, 2)
            del R
            gA = r[0] @ s[0].T + r[0] @ s[1].T + r[1] @ s[0].T
            gS = a[0].T @ r[0] + a[1].T @ r[0] + a[0].T @ r[1]
            return gA * (f32(1) / (sR * sS)), gS * (f32(1) / (sR * sA))
        if balance:
            # per-component powers of two d_k: S row k up, A column k down by the same factor (A S unchanged) so that both carry the
            # component's strength sqrt(max|A_k| max|S_k|); the gradient columns are unscaled per component at the end
            mA, mS = np.abs(A).max(axis=0), np.abs(S).max(axis=1)
            with np.errstate(divide="ignore", invalid="ignore"):
                d = np.where((mA > 0) & (mS > 0), np.rint(0.5 * np.log2(mA / mS)), 0.0)
            dk = np.ldexp(np.ones(A.shape[1]), d.astype(int)).astype(f32)
            gA, gS = make_grad(nP, drop_ll, nR, nG, gmax, False)(A / dk[None, :], S * dk[:, None], Y)
            return gA / dk[None, :], gS * dk[:, None]
        sA, sS = scale_of(np.abs(A).max()), scale_of(np.abs(S).max())
        a = terms16(A * sA, nP)
        s = terms16(S * sS, nP)
        P = np.zeros((A.shape[0], S.shape[1]), f32)
        for i in range(nP):
            for j in range(nP):
                if i + j >= nP and not (i + j == 2 and nP == 2 and not drop_ll):
                    continue
                P += a[i] @ s[j]
        P *= f32(1) / (sA * sS)
        R = P - Y
        del P
        sR = scale_of(np.abs(Y).max() + K * np.abs(A).max() * np.abs(S).max())
        r = terms16(R * sR, nR)
        del R
        if nG > nP:
            a, s = terms16(A * sA, nG), terms16(S * sS, nG)
        a2, s2 = a[:nG], s[:nG]
        gA = np.zeros(A.shape, f32)
        gS = np.zeros(S.shape, f32)
        lim = gmax if gmax is not None else max(nR, nG)
        for i in range(nR):
            for j in range(nG):
                if i + j >= lim:
                    continue
                gA += r[i] @ s2[j].T
                gS += a2[j].T @ r[i]
        out = gA * (f32(1) / (sR * sS)), gS * (f32(1) / (sR * sA))
        assert out[0].dtype == f32 and out[1].dtype == f32
        return out
    return grad


def frac_out(x, ref):
    err = np.abs(x.astype(np.float64) - ref)
    b = 1e-5 + 1e-4 * np.abs(ref)
    return float((err > b).mean()), float((err / b).max())


Y, A0, S0 = orc.synthetic_problem(M, N, K, f32, unity_S=True, seed=4321)
t0 = time.time()
A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
orc.adaprox_nmf(Y.astype(np.float64), A64, S64, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=ITS, e_rel=1e-3, check_convergence=False)
print("fp64 oracle: %.0f s" % (time.time() - t0), flush=True)
real = orc.residual_gradients
base = None
MODES = (("f32", None), ("f16x2", make_grad(2, True, 2)), ("f16r3", make_grad(3, True, 2)),
         ("g: R2 x S2 +ll", make_grad(2, True, 2, 2, 3)),        # the low x low product in the gradient contractions (4 products)
         ("g: R3 x S3 (6)", make_grad(2, True, 3, 3, 3)),        # three terms of both operands there, products down to 2^-33 (6 products)
         ("all3", make_grad(3, True, 3, 3, 3)),
         ("2 acc: hl+lh", make_grad(2, True, 2, split_acc=1)),
         ("2 acc: +h3+3h", make_grad(2, True, 2, split_acc=2)),
         ("2 acc: +ll", make_grad(2, True, 2, split_acc=3)),
         ("f16x2 + Gram fix", make_grad(2, True, 2, gram_fix=1)),
         ("f16x2 + Gram fix 2", make_grad(2, True, 2, gram_fix=2)))
SEL = sys.argv[3] if len(sys.argv) > 3 else None
for name, g in MODES:
    if SEL and name != 'f32' and SEL not in name:
        continue
    orc.residual_gradients = g if g is not None else real
    t0 = time.time()
    A, S = A0.copy(), S0.copy()
    ret = orc.adaprox_nmf(Y, A, S, ("plus",), ("unity_plus", 0), scheme="amsgrad", max_iter=ITS, e_rel=1e-3, check_convergence=False)
    oa, wa = frac_out(A, A64)
    os_, ws = frac_out(S, S64)
    if base is None:
        base = (max(oa, 1.0 / A.size), max(os_, 1.0 / S.size))
    print("%-16s out of tolerance A %.3e (%.1f x yardstick) S %.3e (%.1f x) worst %.0f x the bound; passes %s; %.0f s" % (
        name, oa, oa / base[0], os_, os_ / base[1], max(wa, ws), list(ret[5]), time.time() - t0), flush=True)
orc.residual_gradients = real
