"""3-channel input layers (the 6 -> 64 row shows the general path twice: not dispatched to these kernels): csrc/sphere_conv_small.hip against the general path (im2col + library GEMM + ATen activation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib
from emlight_amd.GenProjector import spherenet
from emlight_amd.GenProjector.spherenet import SphereConv2D


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
    return a.elapsed_time(b) / n


class NoSmall:
    def __init__(self, L):
        self.L = L

    def __getattr__(self, k):
        if k == "eml_sphere_conv_small_supported":
            return lambda c, o: 0
        return getattr(self.L, k)


real = _lib.lib
for (B, C, O, H, W, stride, slope, xgrad) in [(32, 3, 128, 128, 256, 1, 0.0, False), (32, 3, 128, 64, 128, 1, 0.0, False),
                                              (32, 3, 128, 32, 64, 1, 0.0, False), (64, 6, 64, 128, 256, 2, 0.2, True),
                                              (32, 3, 64, 128, 256, 1, 0.0, True)]:
    conv = SphereConv2D(C, O, stride=stride).cuda()
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(xgrad)
    out = {}
    for mode in ("small", "general"):
        _lib.lib = real if mode == "small" else (lambda L=real(): NoSmall(L))
        y = conv(x, act_slope=slope)
        gy = torch.randn_like(y)
        f = timed(lambda: conv(x, act_slope=slope))

        def fb():
            yy = conv(x, act_slope=slope)
            yy.backward(gy)
        out[mode] = (f, timed(fb) - f)
    _lib.lib = real
    gb = B * (H // stride) * (W // stride) * O * 4 / 1e9
    print("B%d %d->%d @%dx%d s%d: output %.2f GB | fwd small %.3f ms (%.0f GB/s) general %.3f | bwd small %.3f general %.3f"
          % (B, C, O, H, W, stride, gb, out["small"][0], gb / out["small"][0] * 1e3, out["general"][0], out["small"][1],
             out["general"][1]))
