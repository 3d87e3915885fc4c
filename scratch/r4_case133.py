# scratch: replay fuzz_nmf3.py 53 250 case 133 and vary its arguments
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
g.build()
import proxmin_amd as pm
from oracle import nmf_oracle as orc
import fuzz_nmf, logging
logging.getLogger("proxmin").setLevel(logging.ERROR)
cap = {}
real = pm.nmf.nmf
def grab(Y, A, S, **kw):
    cap["Y"], cap["A0"], cap["S0"], cap["kw"] = Y.copy(), A.copy(), S.copy(), dict(kw)
    return real(Y, A, S, **kw)
pm.nmf.nmf = grab
fuzz_nmf.run_options(53, 250, only={133}, log=print)
pm.nmf.nmf = real
Y, A0, S0, kw = cap["Y"], cap["A0"], cap["S0"], cap["kw"]
kw.pop("callback", None); kw.pop("algorithm", None)
pA, pS = kw.pop("prox_A"), kw.pop("prox_S")
print({k: (v if np.isscalar(v) else "array[%d] %s.." % (len(v), v[:3])) for k, v in kw.items()}, pA, pS)
def frac(a, b):
    r = np.abs(a.astype(np.float64) - b) / (2e-5 + 2e-4 * np.abs(b))
    return "%.4f/%.1f" % (float((r <= 1).mean()), float(r.max()))
def trial(name, **over):
    k2 = dict(kw); k2.update(over)
    A64, S64 = A0.astype(np.float64), S0.astype(np.float64)
    orc.adaprox_nmf(Y.astype(np.float64), A64, S64, ("plus",), ("plus",), **k2)
    A32, S32 = A0.copy(), S0.copy()
    orc.adaprox_nmf(Y, A32, S32, ("plus",), ("plus",), **k2)
    out = "%-28s fp32 oracle A %s S %s |" % (name, frac(A32, A64), frac(S32, S64))
    for mode in ("f32", "f16x2"):
        pm.set_default_mode(mode)
        A, S = A0.copy(), S0.copy()
        pm.nmf.nmf(Y, A, S, algorithm=pm.adaprox, prox_A=pA, prox_S=pS, **k2)
        out += " device %s A %s S %s |" % (mode, frac(A, A64), frac(S, S64))
    pm.set_default_mode("f32")
    print(out, flush=True)
n = kw["max_iter"]
trial("as found")
trial("e_rel 1e-3", e_rel=1e-3)
trial("check_convergence False", check_convergence=False)
trial("b1 = 0.9", b1=0.9)
trial("b1 const array", b1=np.full(n, 0.9))
trial("12 iterations", max_iter=12, b1=np.asarray(kw["b1"])[:12] if not np.isscalar(kw["b1"]) else kw["b1"])
trial("adam", scheme="adam")
