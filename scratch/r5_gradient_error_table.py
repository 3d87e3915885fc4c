# scratch (CPU, NumPy): which decompositions of K1's three contractions keep the gradients in exact fp32's error class, and what they cost in
# MFMA issue slots.  Gradient level: rms error of gA / gS against fp64, in units of max|g|, at a unity-column problem (the regime of the full-size
# parity runs: R ~ P, gradients coherent).  fp16 / fp8-like terms are held in fp32 arrays (products exact), accumulation is NumPy's fp32 GEMM.
#   python scratch/r5_gradient_error_table.py [M] [N] [K]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import nmf_oracle as orc

f32 = np.float32
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64


def t16(x, n):
    out, r = [], x
    for _ in range(n):
        t = r.astype(np.float16).astype(f32)
        out.append(t)
        r = r - t
    return out


def rnd_bits(x, bits):
    """x rounded to `bits` significant bits (round to nearest even), any exponent: the precision of an fp8 e4m3 / fp6 e2m3 operand (4 bits)
    without their range limits."""
    m, e = np.frexp(x.astype(np.float64))
    return (np.ldexp(np.rint(np.ldexp(m, bits)), e - bits)).astype(f32)


def scale_of(m, top=14):
    q = np.frexp(f32(m))[1]
    return f32(np.ldexp(1.0, int(top - q))) if m > 0 else f32(1)


def grad(A, S, Y, prod="x2", cons="3", lowbits=4):
    """prod: how P = A S is formed  x2 = hh+hl+lh one accumulator (mode f16x2) | r3 = two accumulators + third terms (mode f16x2r)
                                     | hh = high x high ONLY, the rest restored exactly through K x K Gram matrices (mode f16x2g)
                                     | hh_nofix = the same without the correction
       cons: the gradient contractions  3 = r0 s0 + r0 s1 + r1 s0 (today) | 1 = r0 s0 only | 8 = r0 s0 in fp16 + the two cross products with
             both operands rounded to `lowbits` bits (an fp8 / fp6 MFMA)"""
    sA, sS = scale_of(np.abs(A).max()), scale_of(np.abs(S).max())
    a, s = t16(A * sA, 3), t16(S * sS, 3)
    u = f32(1) / (sA * sS)
    if prod == "x2":
        P = a[0] @ s[0]
        P += a[0] @ s[1]
        P += a[1] @ s[0]
        R = P * u - Y
    elif prod == "r3":
        lo = a[0] @ s[1] + a[1] @ s[0] + a[0] @ s[2] + a[2] @ s[0]
        R = ((a[0] @ s[0]) * u - Y) + lo * u
    else:
        R = (a[0] @ s[0]) * u - Y
    sR = scale_of(np.abs(Y).max() + K * np.abs(A).max() * np.abs(S).max())
    r = t16(R * sR, 2)
    if cons == "3":
        gA = r[0] @ s[0].T + r[0] @ s[1].T + r[1] @ s[0].T
        gS = a[0].T @ r[0] + a[1].T @ r[0] + a[0].T @ r[1]
    elif cons == "1":
        gA = r[0] @ s[0].T
        gS = a[0].T @ r[0]
    elif cons == "2r":      # the residual's low term dropped, the factors' kept
        gA = r[0] @ s[0].T + r[0] @ s[1].T
        gS = a[0].T @ r[0] + a[1].T @ r[0]
    elif cons == "2f":      # the factors' low terms dropped, the residual's kept
        gA = r[0] @ s[0].T + r[1] @ s[0].T
        gS = a[0].T @ r[0] + a[0].T @ r[1]
    else:
        q = lambda x: rnd_bits(x, lowbits)
        gA = r[0] @ s[0].T + q(r[0]) @ q(s[1]).T + q(r[1]) @ q(s[0]).T
        gS = a[0].T @ r[0] + q(a[1]).T @ q(r[0]) + q(a[0]).T @ q(r[1])
    gA = gA * (f32(1) / (sR * sS))
    gS = gS * (f32(1) / (sR * sA))
    if prod == "hh":
        # what the high x high product left out, exactly:  A S - a0 s0 = A s_rest + a_rest s0   (a0, s0: the high terms un-scaled; *_rest = X - x0, exact in fp32)
        #   gA += A (s_rest S^T) + a_rest (s0 S^T)        gS += (A^T A) s_rest + (A^T a_rest) s0
        a0, s0 = a[0] / sA, s[0] / sS
        ar, sr = A - a0, S - s0
        gA = gA + A @ (sr @ S.T) + ar @ (s0 @ S.T)
        gS = gS + (A.T @ A) @ sr + (A.T @ ar) @ s0
    return gA.astype(f32), gS.astype(f32)


Y, A, S = orc.synthetic_problem(M, N, K, f32, unity_S=True, seed=4321)
gA64, gS64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))


def err(g, g64):
    return float(np.sqrt(((g - g64) ** 2).mean()) / np.abs(g64).max())


rows = [("numpy fp32 (the yardstick)", None, "-"),
        ("f16x2   P 3 products, gradients 3+3", dict(prod="x2", cons="3"), "36"),
        ("f16x2r  P 5 products / 2 acc, gradients 3+3", dict(prod="r3", cons="3"), "44"),
        ("hh only, NO correction, gradients 3+3", dict(prod="hh_nofix", cons="3"), "28"),
        ("f16x2g  P hh + Gram correction, gradients 3+3", dict(prod="hh", cons="3"), "28"),
        ("f16x2g, gradients r0 s0 only", dict(prod="hh", cons="1"), "12"),
        ("f16x2g, gradients r0 (s0 + s1): residual's low term dropped", dict(prod="hh", cons="2r"), "20"),
        ("f16x2g, gradients (r0 + r1) s0: factors' low terms dropped", dict(prod="hh", cons="2f"), "20"),
        ("f16x2g, gradient cross terms with 4-bit operands", dict(prod="hh", cons="8", lowbits=4), "4 + 8 + (8 fp8 | 4 fp6)"),
        ("f16x2g, gradient cross terms with 3-bit operands", dict(prod="hh", cons="8", lowbits=3), ""),
        ("f16x2g, gradient cross terms with 6-bit operands", dict(prod="hh", cons="8", lowbits=6), ""),
        ("f16x2r, gradient cross terms with 4-bit operands", dict(prod="r3", cons="8", lowbits=4), ""),
        ]
print("# %d x %d x %d, unity columns: rms error of the gradients against fp64 in units of max|g|  (gA / gS)   [fp16 MFMAs per SIMD and slot]" % (M, N, K))
for name, kw, cost in rows:
    if kw is None:
        gA, gS = orc.residual_gradients(A, S, Y)
    else:
        gA, gS = grad(A, S, Y, **kw)
    print("%-52s %.2e / %.2e   [%s]" % (name, err(gA, gA64), err(gS, gS64), cost), flush=True)
