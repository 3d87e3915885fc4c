# scratch: randomized end-to-end sweeps of the fp64 path at size (tests/fuzz_nmf.run(F64="big")): shapes up to 2200 x 3000 x 128, the three back-ends, every operator
import sys, os, logging
os.environ.setdefault("PMX_TORCH_PRELOAD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.build()
import fuzz_nmf
logging.getLogger("proxmin").setLevel(logging.ERROR)
tot = 0
seeds = [int(x) for x in sys.argv[1:]] or [91, 92]
for seed in seeds:
    lines = []
    bad = fuzz_nmf.run(seed, 60, log=lines.append, F64="big")
    tot += bad
    print("run F64=big seed %d: 60 cases, %d bad" % (seed, bad), flush=True)
    for l in lines:
        if not l.startswith("ok"):
            print("   ", l[:300], flush=True)
print("TOTAL bad:", tot)
