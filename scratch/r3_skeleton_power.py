# scratch (round 3): package power / shader clock while the skeletons of scratch/ystream2.hip run ("hold" mode): Y alone,
# K1's MFMAs alone, both -- the energy terms of DESIGN.md section 5.  Output: gpurun_out/r3_power/skeleton.txt
import ctypes, glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "r3_power")
os.makedirs(OUT, exist_ok=True)
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
BDF = buf.value.decode().lower() if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0 else None
BDF = buf.value.decode().lower()
hw = None
for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    if os.path.realpath(os.path.dirname(os.path.dirname(d))).lower().endswith(BDF):
        hw = d
assert hw, "no hwmon for %s" % BDF


def rd(f):
    try:
        return float(open(os.path.join(hw, f)).read())
    except Exception:
        return float("nan")


lines = ["HIP device 0 = PCI %s; hwmon %s; cap %.0f W" % (BDF, hw, rd("power1_cap") / 1e6)]
for variant, secs in (("idle", 3), ("stream4", 5), ("stream_v8", 5), ("mfma", 5), ("mfma_ldsrand", 5), ("both4", 5), ("both_v8", 5),
                      ("both_v8_ldsconst", 5), ("both_v8_ldsrand", 5), ("both_v8_ldsrand_epi", 5)):
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            pw = rd("power1_average")
            if pw != pw:
                pw = rd("power1_input")
            samples.append((pw / 1e6, rd("freq1_input") / 1e6))
            time.sleep(0.02)
    th = threading.Thread(target=sampler)
    th.start()
    if variant == "idle":
        time.sleep(secs); msg = "idle"
    else:
        r = subprocess.run([os.path.join(ROOT, "scratch", "ystream2"), "hold", variant, str(secs)], capture_output=True, text=True)
        msg = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
    stop.set(); th.join()
    mid = samples[len(samples) // 4: -max(1, len(samples) // 10)] or samples      # drop the ramp and the tail
    pw = [s[0] for s in mid]; fq = [s[1] for s in mid]
    ms = float(msg.split(":")[1].split("ms")[0]) if "ms per launch" in msg else float("nan")
    line = "%-20s %-64s power W mean %.0f (min %.0f, max %.0f)  sclk MHz mean %.0f   energy per launch %.3f J (dynamic, above 298 W: %.3f J)" % (
        variant, msg[:64], sum(pw) / len(pw), min(pw), max(pw), sum(fq) / len(fq), sum(pw) / len(pw) * ms * 1e-3, (sum(pw) / len(pw) - 298.0) * ms * 1e-3)
    print(line, flush=True)
    lines.append(line)
open(os.path.join(OUT, "skeleton.txt"), "w").write("\n".join(lines) + "\n")
