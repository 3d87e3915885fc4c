# scratch: is K1 power-limited?  Same data as bench cfg3; K1 inside the iteration chain vs back to back, with power / clock samples
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
device = torch.device("cuda", 0)
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, device)
dev = DeviceNMF(M, N, K, device=0, mode=os.environ.get("PMX_AB_MODE", "bf16x3"))
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
run = bench.begin_solver(dev, backend, unity)
samples = []
stop = [False]
def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append(" | ".join(l.split(":", 1)[-1].strip() for l in out.splitlines() if "sclk" in l or "Power" in l))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.1)
def phase(name, fn, secs=4.0):
    samples.clear(); stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); r = None
    while time.time() - t0 < secs:
        r = fn()
    stop[0] = True; th.join()
    print("%-28s %s" % (name, r)); print("     ", samples[len(samples)//2:][:3], flush=True)
run(20)
def chain():
    dev.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, n = dev.get_timing(); dev.set_timing(False)
    return "iteration %.3f ms, K1 avg %.3f ms" % (dt / 200 * 1e3, ms / max(n, 1))
phase("chain", chain)
for doA, doS in ((1, 1), (0, 0), (1, 0), (0, 1)):
    phase("K1 back-to-back doA=%d doS=%d" % (doA, doS), lambda: "%.3f ms" % dev.time_grad(doA, doS, 400))
phase("chain again", chain)
