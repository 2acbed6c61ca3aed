"""Does the gather's L2 window limit the fused SphereConv kernels?  Forward and input gradient (TF/s) at 128 x 256, B = 32,
for growing row widths of the gathered operand: a window of 64 concurrently running 128-pixel tiles per XCD spans 32 image
rows, i.e. 32 * 256 * C * 4 bytes = 2.1 / 4.2 / 8.4 MB at C = 64 / 128 / 256 against a 4 MB L2.
    python tools/exp/gg_locality.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402

SphereConv2D.fused_min_bytes = 0
B = int(os.environ.get("B", 32))


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for C, O, H, W in [(64, 128, 128, 256), (128, 128, 128, 256), (256, 128, 128, 256), (128, 64, 128, 256), (128, 256, 128, 256),
                   (256, 256, 64, 128), (512, 256, 64, 128), (256, 512, 64, 128)]:
    m = SphereConv2D(C, O, bias=False).cuda()
    m.weight.requires_grad_(False)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.no_grad():
        f_ms = timed(lambda: m(x))
    y = m(x)
    gy = torch.randn_like(y)
    d_ms = timed(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
    fl = 2.0 * B * H * W * 9 * C * O / 1e9
    print("%4d -> %4d @%dx%d  fwd %.3f ms %6.1f TF/s (gathers rows of %4d B) | dgrad %.3f ms %6.1f TF/s (gathers rows of %4d B)"
          % (C, O, H, W, f_ms, fl / f_ms, 4 * C, d_ms, fl / d_ms, 4 * O))
