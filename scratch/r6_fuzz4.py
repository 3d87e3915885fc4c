# scratch: the randomized end-to-end sweeps of tests/fuzz_nmf.py once more on the final build of round 6 (chain prefetch in): new seeds
import sys, os, logging, collections
os.environ.setdefault("PMX_TORCH_PRELOAD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.build()
import fuzz_nmf
logging.getLogger("proxmin").setLevel(logging.ERROR)
tot = 0
for name, fn, seeds, n, kw in (("run", fuzz_nmf.run, (111, 112), 80, {}), ("run_options", fuzz_nmf.run_options, (113,), 60, {}), ("run F64", fuzz_nmf.run, (114,), 40, {"F64": True})):
    for seed in seeds:
        lines = []
        bad = fn(seed, n, log=lines.append, **kw)
        tot += bad
        modes = collections.Counter(w for l in lines for w in l.split() if w in ("f32", "bf16x3", "f16x2", "f16x2r"))
        print("%s seed %d: %d cases, %d bad; modes drawn %s" % (name, seed, n, bad, dict(modes)), flush=True)
        for l in lines:
            if not l.startswith("ok"):
                print("   ", l[:300], flush=True)
print("TOTAL bad:", tot)
