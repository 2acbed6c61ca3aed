"""Does a kernel run at full clock?  Loops one conv3x3 fwd launch for a few seconds while sampling the GPU's
sclk / power from hwmon (sysfs), and prints ms per launch + the clock/power distribution."""
import sys, os, glob, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd import _lib
if os.environ.get('EML_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EML_LIB_PATH']
L, p = _lib.lib(), _lib.ptr
st = _lib.current_stream()
B, Hb, Wb, Cin, ld = 64, 240, 320, 204, 224
P = B * Hb * Wb
dev = "cuda"
X = torch.randn(P, ld, device=dev); Z = torch.randn(P, 48, device=dev)
s2 = torch.rand(48, device=dev) + 0.5; t2 = torch.randn(48, device=dev) * 0.2
W2 = torch.randn(12, 48, 3, 3, device=dev) * 0.1; W2p = torch.empty(6912, device=dev)
L.eml_dense_permute_w2_f32(p(W2), 12, p(W2p), st)
part = torch.zeros(4 * 1024 * 96, dtype=torch.float64, device=dev)
fn = lambda: L.eml_dense_conv3x3_fwd_f32(p(Z), p(s2), p(t2), p(W2p), p(X), ld, Cin, B, Hb, Wb, p(part), 512, st)
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
samples, stop = [], False
def rd(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None
def sampler():
    while not stop:
        row = []
        for h in hw:
            row.append((rd(h + "/freq1_input"), rd(h + "/power1_average") or rd(h + "/power1_input")))
        samples.append(row)
        time.sleep(0.05)
fn(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print("ms/launch %.4f over %d launches" % (e0.elapsed_time(e1) / n, n))
for i, h in enumerate(hw):
    f = [r[i][0] for r in samples if r[i][0]]; w = [r[i][1] for r in samples if r[i][1]]
    if f:
        print(h, "sclk MHz min/mean/max %.0f %.0f %.0f" % (min(f) / 1e6, sum(f) / len(f) / 1e6, max(f) / 1e6),
              "power W min/mean/max %.0f %.0f %.0f" % ((min(w) / 1e6, sum(w) / len(w) / 1e6, max(w) / 1e6) if w else (0, 0, 0)))
