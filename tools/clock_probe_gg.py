"""Is the gather-GEMM's rate set by cycles or by the clock?  Each variant loops for a few seconds while the GPU's sclk and board
power are sampled from hwmon: the 16-load gather, the 10-load (row-shared) gather, the same kernel on a table whose every tap
is pixel 0 (loads that cannot miss), a library f32 GEMM of the same FLOPs, and the encoder-free baseline of an idle chip.
    python tools/clock_probe_gg.py [seconds per variant] > profiles/r05_clock_probe.txt"""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib
from emlight_amd.GenProjector import spherenet

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda")
L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")


def rd(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None


def run(name, fn, gflop):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append([(rd(h + "/freq1_input"), rd(h + "/power1_average") or rd(h + "/power1_input")) for h in hw])
            time.sleep(0.02)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    # a node exposes one hwmon per GPU: report the one that is actually clocking up (ours), not the average over idle neighbours
    best, mhz, fmin, watts = None, float("nan"), 0.0, float("nan")
    for i in range(len(hw)):
        f = [r[i][0] for r in samples if r[i][0]]
        if f and (best is None or sum(f) / len(f) > best):
            best = sum(f) / len(f)
            w = [r[i][1] for r in samples if r[i][1]]
            mhz, fmin, watts = best / 1e6, min(f) / 1e6, (sum(w) / len(w) / 1e6 if w else float("nan"))
    tf = gflop / ms
    print("%-52s %7.4f ms  %6.1f TF/s  sclk %5.0f MHz (min %4.0f)  %4.0f W  -> %.1f %% of the f32-MFMA peak AT THAT CLOCK"
          % (name, ms, tf, mhz, fmin, watts, 100 * tf / (157.3 * mhz / 2400.0)), flush=True)


B, C, O, H, W = 32, 128, 128, 128, 256
geo = spherenet.sphere_geometry(H, W, 1, dev)
po, M = H * W, B * H * W
gflop = 2.0 * M * 9 * C * O / 1e9
x = torch.randn(M, C, device=dev)
w2 = torch.randn(O, 9 * C, device=dev) * 0.05
y = torch.empty(M, O, device=dev)
idx0 = torch.zeros_like(geo.idx)


def conv(idx, flags):
    return lambda: _lib.check(L.eml_sphere_conv_fwd_fused_ex_f32(p(x), p(idx), p(geo.wgt), p(w2), None, p(y), B, po, po, C, O, 4, None,
                                                                 1.0, flags, st), "fwd")


print("layer 128 -> 128 @128x256, B = 32 (%.1f GFLOP); %d hwmon node(s)" % (gflop, len(hw)))
run("gather-GEMM, 16 gathered loads per chunk", conv(geo.idx, 0), gflop)
run("gather-GEMM, 10 (row-shared corners)", conv(geo.idx, 1), gflop)
run("gather-GEMM, every tap -> pixel 0 (loads cannot miss)", conv(idx0, 0), gflop)
a9 = torch.randn(M, 9 * C, device=dev)
wt = w2.t().contiguous()
run("library f32 GEMM (M x 1152) @ (1152 x 128)", lambda: torch.mm(a9, wt, out=y), gflop)
xz = torch.zeros_like(x)
run("gather-GEMM, 16 loads, all-zero activations", lambda: _lib.check(L.eml_sphere_conv_fwd_fused_ex_f32(
    p(xz), p(geo.idx), p(geo.wgt), p(w2), None, p(y), B, po, po, C, O, 4, None, 1.0, 0, st), "fwd"), gflop)
# a bigger library GEMM at its best shape, for the clock a saturated matrix pipe settles at
a = torch.randn(8192, 8192, device=dev)
b = torch.randn(8192, 8192, device=dev)
c = torch.empty(8192, 8192, device=dev)
run("library f32 GEMM 8192^3", lambda: torch.mm(a, b, out=c), 2.0 * 8192 ** 3 / 1e9)
