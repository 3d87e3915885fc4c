# scratch (round 3): runtime ablation switches of k_grad_f16_v8 (doA bits 8+; TEMPORARY build), cycles per slot from
# time x sampled shader clock.  flags: 1 no Y loads, 2 no producer MFMAs, 4 no epilogue (VALU + R stores), 8 no S fragment
# reads, 16 no R stores (VALU kept), 32 consumers' gA operand reads off, 64 consumers' gSt operand reads off
import sys, os, glob, threading, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PMX_K1_CHAIN", os.environ.get("CHAIN", "1"))
import numpy as np, torch
import __graft_entry__ as g
g.build()
import bench
from proxmin_amd.engine import DeviceNMF

def my_hwmon():
    hip = ctypes.CDLL("libamdhip64.so"); buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0: return None
    bdf = buf.value.decode().lower()
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.realpath(os.path.dirname(os.path.dirname(d))).lower().endswith(bdf): return d
HW = my_hwmon()
def rd(f):
    try: return float(open(HW + "/" + f).read())
    except Exception: return float("nan")
N, K = 16384, 64
device = torch.device("cuda", 0)
Yfull, A0f, S0f = bench.make_problem_device(16384, N, K, True, 1234, device)
devs = {}
for M in (16384, 8192):
    d = DeviceNMF(M, N, K, device=0, mode="f16x2")
    d.set_Y_device(Yfull.data_ptr(), ld=N, copy=False, keepalive=Yfull)
    d.set_factors(A0f[:M], S0f)
    devs[M] = d
    print(M, d.k1_info())
def measure(dev, doA, doS, secs=1.0):
    samples, stop = [], [False]
    def smp():
        while not stop[0]:
            samples.append((rd("freq1_input") * 1e-6, rd("power1_average") * 1e-6 if os.path.exists(HW + "/power1_average") else rd("power1_input") * 1e-6)); time.sleep(0.02)
    th = threading.Thread(target=smp); th.start()
    t0 = time.time(); ms = []
    while time.time() - t0 < secs: ms.append(dev.time_grad(doA, doS, 200))
    stop[0] = True; th.join()
    s = samples[len(samples) // 3:]
    f = sum(x[0] for x in s) / len(s); p = sum(x[1] for x in s) / len(s)
    return ms[-1], f, p
cases = [("full", 0), ("empty producer", 15), ("empty producer, no A split/publish", 15 + 128), ("no A split/publish", 128),
         ("consumers reads off + empty producer + no A", 96 + 15 + 128)]
for doA, doS in ((0, 0), (1, 1)):
    for name, fl in cases:
        if doA == 0 and fl >= 32: continue
        r = {}
        for M in (16384, 8192):
            r[M] = measure(devs[M], doA | (fl << 8), doS)
        dt = (r[16384][0] - r[8192][0]) * 1e-3; f = 0.5 * (r[16384][1] + r[8192][1]) * 1e6
        fixed = 2 * r[8192][0] - r[16384][0]
        print("doA=%d doS=%d %-44s %.4f / %.4f ms  %4.0f MHz %4.0f W -> slope %5.0f cycles/slot (%.3f us), fixed %.1f us per launch" % (
            doA, doS, name, r[16384][0], r[8192][0], r[16384][1], r[16384][2], dt * f / 128.0, dt / 128 * 1e6, fixed * 1e3), flush=True)
