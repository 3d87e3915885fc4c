"""CPU: a NumPy model of the device's split-fp16 products (fp16 terms held in fp32 arrays: products exact, fp32 accumulate) that shows, without
a GPU, WHY mode f16x2's gradients are twice as far from fp64 as exact fp32's and why mode f16x2r's are not (k_grad_f16_v8.hip <R3>, DESIGN
section 2; scratch/r4_emulate_modes.py runs the whole solver on this model at full size: profiles/r04_k_emulated_modes.txt).

With two fp16 terms per operand the residual P - Y is computed from operands rounded to 2^-23: the error dS[k][n] of an entry of S enters every
row of P with the same sign, and gS = A^T R sums it over the rows -- (A^T A) dS, coherent, where rounding errors add up like sqrt(M).  More terms
in the SAME accumulator do not help: their products lie below half an ulp of the accumulated P.  The third terms in a SECOND accumulator,
R = (P_hh - Y) + P_lo, do.  The GPU test of the same statement: tests/test_gpu_kernels.py::test_third_terms_in_the_residual_remove_the_coherent_error."""
import numpy as np

from oracle import nmf_oracle as orc

f32 = np.float32


def _terms(x, n):
    out, r = [], x.astype(f32)
    for _ in range(n):
        t = r.astype(np.float16).astype(f32)
        out.append(t)
        r = r - t
    return out


def _scale(m, top=14):
    return f32(np.ldexp(1.0, int(top - np.frexp(f32(m))[1])))


def _gS(A, S, Y, n_terms, second_accumulator):
    """gS = A^T (A S - Y) with the device's arithmetic for A S: n_terms fp16 terms per operand, products down to 2^-22 (2^-33 with three
    terms) of the largest, in one fp32 accumulator or with everything but high x high in a second one; the gradient contraction itself in
    plain fp32 (it is not where the effect lives)."""
    sA, sS = _scale(np.abs(A).max()), _scale(np.abs(S).max())
    a, s = _terms(A * sA, n_terms), _terms(S * sS, n_terms)
    u = f32(1) / (sA * sS)
    pairs = [(i, j) for i in range(n_terms) for j in range(n_terms) if 0 < i + j < n_terms]
    if second_accumulator:
        lo = sum(a[i] @ s[j] for i, j in pairs)
        R = ((a[0] @ s[0]) * u - Y) + lo * u
    else:
        P = a[0] @ s[0]
        for i, j in pairs:
            P += a[i] @ s[j]
        R = P * u - Y
    return A.T @ R


def test_coherent_representation_error_and_the_second_accumulator():
    M = N = 1024
    K = 64
    Y, A, S = orc.synthetic_problem(M, N, K, f32, unity_S=True, seed=4321)
    g64 = orc.residual_gradients(A.astype(np.float64), S.astype(np.float64), Y.astype(np.float64))[1]

    def err(g):
        return float(np.sqrt(((g - g64) ** 2).mean()) / np.abs(g64).max())
    e_f32 = err(orc.residual_gradients(A, S, Y)[1])                  # the reference's arithmetic for fp32 inputs (nmf.py:28-41)
    e_2 = err(_gS(A, S, Y, 2, False))                                 # mode f16x2
    e_3_one = err(_gS(A, S, Y, 3, False))                             # a third term in the same accumulator
    e_2_two = err(_gS(A, S, Y, 2, True))                              # a second accumulator without the third terms
    e_3_two = err(_gS(A, S, Y, 3, True))                              # mode f16x2r
    assert e_2 > 1.5 * e_f32, (e_2, e_f32)                            # two terms: the coherent part dominates
    assert e_3_one > 0.9 * e_2 and e_2_two > 0.9 * e_2, (e_3_one, e_2_two, e_2)      # neither ingredient alone changes it
    assert e_3_two < 1.15 * e_f32, (e_3_two, e_f32)                   # both together: exact fp32's error
    # the representation error alone (the same two-term products accumulated in fp64) is what is left over fp32's own noise
    sA, sS = _scale(np.abs(A).max()), _scale(np.abs(S).max())
    a, s = _terms(A * sA, 2), _terms(S * sS, 2)
    P64 = sum(a[i].astype(np.float64) @ s[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0))) / (float(sA) * float(sS))
    e_repr = err((A.astype(np.float64).T @ (P64 - Y)).astype(f32))
    assert e_repr > 0.7 * e_2, (e_repr, e_2)
