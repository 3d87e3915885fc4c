# scratch (round 5): package power and shader clock while the bench's iteration chain runs, per arithmetic mode (mode f16x2r issues 28 fp16 MFMAs per SIMD and
# slot, mode f16x2 36, <R3> 44, exact fp32 its own pipe): hwmon sysfs samples every 20 ms.  Output: gpurun_out/r5_power/summary.txt
import sys, os, threading, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF
OUT = os.path.join(ROOT, "gpurun_out", "r5_power")
os.makedirs(OUT, exist_ok=True)


def my_bdf():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    return buf.value.decode().lower() if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0 else None


BDF = my_bdf()
cands = [(d, {os.path.basename(f) for f in glob.glob(d + "/*")}, os.path.realpath(os.path.dirname(os.path.dirname(d)))) for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")]
mine = [c for c in cands if BDF and c[2].lower().endswith(BDF)] or cands
HW = mine[0] if mine else None


def sample():
    rec = {}
    if HW:
        for f in ("power1_average", "power1_input", "freq1_input", "power1_cap"):
            if f in HW[1]:
                try:
                    rec[f] = float(open(HW[0] + "/" + f).read().strip())
                except Exception:
                    pass
    return rec


lines = []


def phase(name, fn, secs=4.0):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append(sample())
            time.sleep(0.02)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    r = None
    while time.time() - t0 < secs:
        r = fn()
    stop[0] = True
    th.join()
    mid = samples[len(samples) // 4:]

    def stat(keys, scale):
        for k in keys:
            v = [s[k] * scale for s in mid if k in s]
            if v:
                return "%.0f / %.0f / %.0f" % (min(v), sum(v) / len(v), max(v))
        return "n/a"
    line = "%-44s %-44s | power W (min / mean / max) %s | sclk MHz %s | cap W %s" % (name, r, stat(("power1_average", "power1_input"), 1e-6), stat(("freq1_input",), 1e-6), stat(("power1_cap",), 1e-6))
    print(line, flush=True)
    lines.append(line)


phase("idle", lambda: time.sleep(0.2) or "idle", 2.0)
# [r5, later] specs on the command line: cfg:mode[:ENV=V,...]  (default: the four arithmetic modes at cfg3)
specs = [a.split(":") for a in sys.argv[1:]] or [["cfg3", "f16x2r"], ["cfg3", "f16x2"], ["cfg3", "f16x2r", "PMX_F16_R3=1"], ["cfg3", "f32"]]
cache = {}
for sp in specs:
    cfg, mode = sp[0], sp[1]
    env = dict(kv.split("=") for kv in sp[2].split(",")) if len(sp) > 2 else {}
    M, N, K, backend, unity, _ = bench.CONFIGS[cfg]
    if cfg not in cache:
        cache.clear()
        cache[cfg] = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
    Y, A0, S0 = cache[cfg]
    os.environ.update(env)
    dev = DeviceNMF(M, N, K, device=0, mode=mode)
    dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
    dev.set_factors(A0, S0)
    run = bench.begin_solver(dev, backend, unity)
    run(30)
    nit = 200 if M * N >= (1 << 26) else 2000

    def chain():
        dev.set_timing(True, every=4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(nit)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, n = dev.get_timing(); dev.set_timing(False)
        return "iteration %.4f ms, K1 %.4f ms (%s)" % (dt / nit * 1e3, ms / max(n, 1), dev.k1_info()["kernel"].replace("k_grad_", ""))
    phase("%s chain, mode %s %s" % (cfg, mode, ",".join("%s=%s" % kv for kv in env.items())), chain)
    dev.close()
    for k in env:
        del os.environ[k]
open(os.path.join(OUT, "summary.txt"), "w").write("\n".join(lines) + "\n")
