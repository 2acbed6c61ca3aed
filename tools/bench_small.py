"""The 3 -> 128 input layers (SPADE's mlp_shared, 18 per generator pass) kernel by kernel at the generator's seven resolutions,
B = 32: forward, weight gradient (+ bias, + ReLU backward), input gradient's first half -- C entries called directly, HIP
events, back-to-back launches.  EML_LIB_PATH selects an experiment build.     python tools/bench_small.py [O]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib  # noqa: E402
if os.environ.get("EML_LIB_PATH"):
    _lib.LIB_PATH = os.environ["EML_LIB_PATH"]
from emlight_amd.GenProjector.spherenet import SphereGeometry  # noqa: E402

L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
B, C = 32, 3
O = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = [0.0, 0.0, 0.0]
for H, W, count in [(4, 8, 3), (8, 16, 6), (16, 32, 3), (32, 64, 3), (64, 128, 3), (128, 256, 3)]:
    geo = SphereGeometry(H, W, 1, "cuda")
    po = H * W
    M = B * po
    x = torch.randn(M, C, device="cuda")
    w2 = torch.randn(O, 9 * C, device="cuda") * 0.2
    bias = torch.randn(O, device="cuda") * 0.1
    y = torch.empty(M, O, device="cuda")
    gy = torch.randn(M, O, device="cuda")
    part = torch.empty(L.eml_sphere_conv_small_wgrad_partial_floats(B, po, C, O), device="cuda")
    gw2, gb = torch.empty(O, 9 * C, device="cuda"), torch.empty(O, device="cuda")
    da9 = torch.empty(M, 9 * C, device="cuda")
    f = timed(lambda: _lib.check(L.eml_sphere_conv_small_fwd_f32(p(x), p(geo.idx), p(geo.wgt), p(w2), p(bias), p(y), B, po, po, C, O,
                                                                 0.0, st), "fwd"))
    g = timed(lambda: _lib.check(L.eml_sphere_conv_small_wgrad_f32(p(x), p(geo.idx), p(geo.wgt), p(gy), p(y), 0.0, p(part), p(gw2),
                                                                   p(gb), B, po, po, C, O, st), "wgrad"))
    d = timed(lambda: _lib.check(L.eml_sphere_conv_small_da9_f32(p(gy), p(y), 0.0, p(w2), p(da9), M, C, O, st), "da9"))
    out_gb = M * O * 4 / 1e9
    print("%3dx%-3d x%d  fwd %7.1f us (%5.2f TB/s written)  wgrad %7.1f us (%5.2f TB/s read)  da9 %7.1f us   |gw| %.5e"
          % (H, W, count, f, out_gb / f * 1e3, g, 2 * out_gb / g * 1e3, d, float(gw2.double().norm())))
    tot[0] += count * f
    tot[1] += count * g
    tot[2] += count * d
print("per generator pass (18 layers + 3 more at 8x16): fwd %.2f ms  wgrad %.2f ms  da9 %.2f ms" % tuple(t / 1e3 for t in tot))
