"""The few-output-channel layers (conv_img 64 -> 3 at 128 x 256, the discriminators' heads) kernel by kernel, B = 32 / 64: C entries
called directly, HIP events.  EML_LIB_PATH selects an experiment build.     python tools/bench_narrow.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib  # noqa: E402
if os.environ.get("EML_LIB_PATH"):
    _lib.LIB_PATH = os.environ["EML_LIB_PATH"]
from emlight_amd.GenProjector.spherenet import SphereGeometry  # noqa: E402

L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()


def timed(fn, n=10):
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


for B, C, O, H, W in [(32, 64, 3, 128, 256), (64, 512, 1, 15, 31), (64, 512, 1, 7, 15)]:
    geo = SphereGeometry(H, W, 1, "cuda")
    po = geo.ho * geo.wo
    M = B * po
    x = torch.randn(B * H * W, C, device="cuda")
    w2 = torch.randn(O, 9 * C, device="cuda") * 0.1
    bias = torch.randn(O, device="cuda") * 0.1
    y = torch.empty(M, O, device="cuda")
    gy = torch.randn(M, O, device="cuda")
    gx = torch.empty(B * H * W, C, device="cuda")
    part = torch.empty(L.eml_sphere_conv_narrow_wgrad_partial_floats(B, po, C, O), device="cuda")
    gw2 = torch.empty(O, 9 * C, device="cuda")
    tidx, twgt, rowmax, ke = geo.transposed_table()
    f = timed(lambda: _lib.check(L.eml_sphere_conv_narrow_fwd_f32(p(x), p(geo.idx), p(geo.wgt), p(w2), p(bias), p(y), B, H * W, po, C,
                                                                  O, st), "fwd"))
    g = timed(lambda: _lib.check(L.eml_sphere_conv_narrow_wgrad_f32(p(x), p(geo.idx), p(geo.wgt), p(gy), p(part), p(gw2), B, H * W,
                                                                    po, C, O, st), "wgrad"))
    d = timed(lambda: _lib.check(L.eml_sphere_conv_narrow_dgrad_f32(p(gy), p(tidx), p(twgt), ke, p(w2), p(gx), B, H * W, po, C, O,
                                                                    st), "dgrad"))
    scr = torch.empty(L.eml_sphere_conv_narrow_scratch_floats(B, H * W), device="cuda")
    part2 = torch.empty(L.eml_sphere_conv_narrow_wgrad2_partial_floats(B, H * W, C), device="cuda")
    y2, gw3 = torch.empty_like(y), torch.empty_like(gw2)
    f2 = timed(lambda: _lib.check(L.eml_sphere_conv_narrow_fwd2_f32(p(x), p(geo.idx), p(geo.wgt), p(w2), p(bias), p(y2), p(scr), B,
                                                                    H * W, po, C, O, st), "fwd2"))
    g2 = timed(lambda: _lib.check(L.eml_sphere_conv_narrow_wgrad2_f32(p(x), p(tidx), p(twgt), ke, p(rowmax) if ke == 8 else None, p(gy), p(scr), p(part2), p(gw3), B,
                                                                      H * W, po, C, O, st), "wgrad2"))
    gx2 = torch.empty_like(gx)
    d2 = timed(lambda: _lib.check(L.eml_sphere_conv_narrow_dgrad2_f32(p(gy), p(tidx), p(twgt), ke, p(rowmax) if ke == 8 else None, p(w2),
                                                                      p(gx2), p(scr), 1, B, H * W, po, C, O, st), "dgrad2"))
    ex = float((gx2 - gx).abs().max() / gx.abs().max())
    ey = float((y2 - y).abs().max() / y.abs().max())
    ew = float((gw3 - gw2).abs().max() / gw2.abs().max())
    print("B%d %d -> %d @%dx%d: fwd %7.1f -> %7.1f us  wgrad %7.1f -> %7.1f us  dgrad %7.1f -> %7.1f us (V reused)   (x = %.0f MB; max rel diff y %.1e, dW %.1e, dX %.1e)"
          % (B, C, O, H, W, f, f2, g, g2, d, d2, x.numel() * 4 / 1e6, ey, ew, ex))
