# scratch (round 3): power / shader-clock trace while K1 runs, as a committed artefact (task 1a).
# Phases: idle; bench.py's own iteration chain; K1 back to back (full, producers only); the same K1 on all-zero data
# (DVFS check: same instruction stream, no toggling); each phase samples the hwmon sysfs files (fast) and takes one
# `amd-smi metric` snapshot in the middle.  Output: gpurun_out/r3_power/{trace.json,summary.txt}
import sys, os, subprocess, threading, time, json, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF

OUT = os.path.join(ROOT, "gpurun_out", "r3_power")
os.makedirs(OUT, exist_ok=True)
mode = os.environ.get("PMX_AB_MODE", "f16x2")


def my_bdf():
    """PCI address of HIP device 0 (the box's sysfs shows every GPU of the host; only one is ours)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0:
        return None
    return buf.value.decode().lower()


BDF = my_bdf()


def find_hwmon():
    cands = []
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        files = {os.path.basename(f) for f in glob.glob(d + "/*")}
        real = os.path.realpath(os.path.dirname(os.path.dirname(d)))
        cands.append((d, files, real))
    mine = [c for c in cands if BDF and c[2].lower().endswith(BDF)]
    return [(d, f) for d, f, _ in (mine or cands)]


HW = find_hwmon()


def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


def sample_sysfs():
    rec = {"t": time.time()}
    for d, files in HW[:1]:
        for f in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "power1_cap"):
            if f in files:
                rec[f] = read(d + "/" + f)
        dev = os.path.dirname(os.path.dirname(d))
        for f in ("pp_dpm_sclk", "pp_dpm_mclk"):
            v = read(dev + "/" + f)
            if v is not None:
                cur = [l for l in v.splitlines() if l.endswith("*")]
                rec[f] = cur[0] if cur else v.replace("\n", " | ")
    return rec


def smi_snapshot():
    out = {}
    for name, cmd in (("amd-smi", ["amd-smi", "metric", "--power", "--clock", "--csv"]),
                      ("amd-smi-list", ["amd-smi", "list", "--csv"])):
        try:
            out[name] = subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout[-6000:]
        except Exception as e:
            out[name] = repr(e)
    return out


trace = {"hwmon": [(d, sorted(f)) for d, f in HW], "phases": []}
M, N, K, backend, unity, desc = bench.CONFIGS["cfg3"]
device = torch.device("cuda", 0)
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, device)
dev = DeviceNMF(M, N, K, device=0, mode=mode)
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
run = bench.begin_solver(dev, backend, unity)


def phase(name, fn, secs=5.0):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append(sample_sysfs())
            time.sleep(0.02)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    r = None
    snap = None
    while time.time() - t0 < secs:
        r = fn()
        if snap is None and time.time() - t0 > secs / 2:
            t1 = threading.Thread(target=lambda: trace.setdefault("snap_" + name, smi_snapshot()))
            t1.start()
            snap = t1
    stop[0] = True
    th.join()
    if snap is not None:
        snap.join()
    mid = samples[len(samples) // 4:]

    def stat(key, scale):
        v = [float(s[key]) * scale for s in mid if s.get(key) not in (None, "")]
        return (min(v), sum(v) / len(v), max(v)) if v else None

    rec = {"name": name, "result": r, "n_samples": len(samples),
           "power_W(min,mean,max)": stat("power1_average", 1e-6) or stat("power1_input", 1e-6),
           "sclk_MHz(min,mean,max)": stat("freq1_input", 1e-6),
           "cap_W": stat("power1_cap", 1e-6), "pp_dpm_sclk": mid[len(mid) // 2].get("pp_dpm_sclk") if mid else None,
           "samples": samples[::5]}
    trace["phases"].append(rec)
    print("%-34s %s | power W %s | sclk MHz %s | %s" % (name, r, rec["power_W(min,mean,max)"], rec["sclk_MHz(min,mean,max)"], rec["pp_dpm_sclk"]), flush=True)


phase("idle", lambda: time.sleep(0.2) or "idle", 2.0)
run(20)


def chain():
    dev.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, n = dev.get_timing(); dev.set_timing(False)
    return "iteration %.4f ms, K1 avg %.4f ms" % (dt / 200 * 1e3, ms / max(n, 1))


phase("bench chain (K1 + tail)", chain)
for doA, doS in ((1, 1), (0, 0), (1, 0), (0, 1)):
    phase("K1 back-to-back doA=%d doS=%d" % (doA, doS), lambda: "%.4f ms" % dev.time_grad(doA, doS, 400))
# the same kernel on all-zero data
Yz = torch.zeros((M, N), device=device, dtype=torch.float32)
devz = DeviceNMF(M, N, K, device=0, mode=mode)
devz.set_Y_device(Yz.data_ptr(), ld=N, copy=False, keepalive=Yz)
devz.set_factors(np.zeros_like(A0), np.zeros_like(S0))
phase("K1 back-to-back, all-zero data", lambda: "%.4f ms" % devz.time_grad(1, 1, 400))
phase("K1 producers only, all-zero data", lambda: "%.4f ms" % devz.time_grad(0, 0, 400))
# constant (non-zero) data: same toggling story, but non-trivial operands
Yc = torch.full((M, N), 0.37, device=device, dtype=torch.float32)
devc = DeviceNMF(M, N, K, device=0, mode=mode)
devc.set_Y_device(Yc.data_ptr(), ld=N, copy=False, keepalive=Yc)
devc.set_factors(np.full_like(A0, 0.25), np.full_like(S0, 0.125))
phase("K1 back-to-back, constant data", lambda: "%.4f ms" % devc.time_grad(1, 1, 400))
# the same data with a row pitch that is not a power of two (HBM channel / bank mapping check): ld = N + 64 floats
Yp = torch.empty((M, N + 64), device=device, dtype=torch.float32)
Yp[:, :N] = Y
devp = DeviceNMF(M, N, K, device=0, mode=mode)
devp.set_Y_device(Yp.data_ptr(), ld=N + 64, copy=False, keepalive=Yp)
devp.set_factors(A0, S0)
phase("K1 back-to-back, pitch N+64", lambda: "%.4f ms" % devp.time_grad(1, 1, 400))
phase("K1 producers only, pitch N+64", lambda: "%.4f ms" % devp.time_grad(0, 0, 400))
phase("bench chain again", chain)
json.dump(trace, open(os.path.join(OUT, "trace.json"), "w"), indent=0)
with open(os.path.join(OUT, "summary.txt"), "w") as f:
    f.write("mode %s; cfg3 16384 x 16384 x 64; HIP device 0 = PCI %s; hwmon %s\n" % (mode, BDF, [d for d, _ in HW]))
    for p in trace["phases"]:
        f.write("%-36s %-44s power W (min, mean, max) %s  sclk MHz %s  cap W %s  %s\n" % (
            p["name"], p["result"], p["power_W(min,mean,max)"], p["sclk_MHz(min,mean,max)"], p["cap_W"], p["pp_dpm_sclk"]))
    for k in trace:
        if k.startswith("snap_"):
            f.write("\n==== %s ====\n" % k)
            for tool, txt in trace[k].items():
                f.write("-- %s --\n%s\n" % (tool, txt))
print("written", OUT)
