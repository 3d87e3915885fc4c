# scratch (round 4): K1 back to back at cfg3 for ~3 s under a power / sclk sampler (hwmon of the HIP device's own PCI function),
# for whichever library PMX_LIB names: one line "tag ms W GHz".  Used by scratch/r4_v8_abl.sh for the ablation A/B tables.
import sys, os, threading, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from proxmin_amd.engine import DeviceNMF


def my_bdf():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    return buf.value.decode().lower() if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0 else None


BDF = my_bdf()
cands = []
for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    real = os.path.realpath(os.path.dirname(os.path.dirname(d)))
    cands.append((d, real))
HW = ([d for d, r in cands if BDF and r.lower().endswith(BDF)] or [d for d, _ in cands])[:1]


def rd(f):
    try:
        return float(open(os.path.join(HW[0], f)).read())
    except Exception:
        return None


M, N, K, backend, unity, desc = bench.CONFIGS[os.environ.get("CFG", "cfg3")]
if os.environ.get("ROWS"):
    M = int(os.environ["ROWS"])          # (cfg4's 8192-row share)
Y, A0, S0 = bench.make_problem_device(M, N, K, unity, 1234, torch.device("cuda", 0))
dev = DeviceNMF(M, N, K, device=0, mode="f16x2")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
dev.time_grad(1, 1, 50)
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        samples.append((rd("power1_average") or rd("power1_input"), rd("freq1_input")))
        time.sleep(0.02)


th = threading.Thread(target=sampler)
th.start()
t0, ms = time.time(), []
while time.time() - t0 < 3.0:
    ms.append(dev.time_grad(int(os.environ.get("DOA", "1")), int(os.environ.get("DOS", "1")), 200))
stop[0] = True
th.join()
mid = samples[len(samples) // 4:]
pw = [p for p, f in mid if p]
fq = [f for p, f in mid if f]
print("%-8s K1 b2b %.4f ms (min %.4f) | %.0f W | %.3f GHz | %s" % (os.environ.get("TAG", "?"), sum(ms) / len(ms), min(ms), sum(pw) / max(len(pw), 1) * 1e-6,
                                                                  sum(fq) / max(len(fq), 1) * 1e-9, dev.k1_info()["kernel"]), flush=True)
dev.close()
