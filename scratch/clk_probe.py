# scratch: sample the shader clock / power while K1 runs back to back
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from proxmin_amd.engine import DeviceNMF
M = N = 16384; K = 64
Y = torch.rand((M, N), device="cuda")
rng = np.random.default_rng(0)
A0 = rng.random((M, K), dtype=np.float32); S0 = rng.random((K, N), dtype=np.float32)
dev = DeviceNMF(M, N, K, mode=sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
dev.set_Y_device(Y.data_ptr(), ld=N, copy=False, keepalive=Y)
dev.set_factors(A0, S0)
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.2)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
while time.time() - t0 < 6.0:
    ms = dev.time_grad(1, 1, 200)
stop = True; th.join()
print("K1 avg ms:", ms)
for s in samples[:3] + samples[-4:]:
    print(s)
