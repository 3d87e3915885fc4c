# scratch: CLI of the argument sweep in tests/fuzz_nmf.py (run_options).   usage: fuzz_nmf3.py seed n_cases [case,case,...]
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging
import __graft_entry__ as g
g.build()
import fuzz_nmf
logging.getLogger("proxmin").setLevel(logging.ERROR)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
only = set(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
bad = fuzz_nmf.run_options(seed, n_cases, only, log=lambda *a: print(*a, flush=True))
print("fuzz3 done: %d cases, %d failures" % (n_cases, bad), flush=True)
