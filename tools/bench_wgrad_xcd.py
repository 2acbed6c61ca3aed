"""Weight gradient of the fused SphereConv at the projector's high-resolution shapes, B = 32: ms per call (HIP events).
EML_WG_XCD=0/1 and EML_WGRAD_WGS are read from the environment (A/B of the tile order).
    python tools/bench_wgrad_xcd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402

SphereConv2D.fused_min_bytes = 0
B = 32
for C, O, H, W in [(128, 256, 128, 256), (128, 128, 128, 256), (128, 64, 128, 256), (64, 64, 128, 256), (128, 512, 64, 128),
                   (256, 128, 64, 128), (128, 128, 64, 128), (512, 256, 32, 64)]:
    m = SphereConv2D(C, O).cuda()
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    y = m(x)
    gy = torch.randn_like(y)
    (gw,) = torch.autograd.grad(y, m.weight, gy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        (gw,) = torch.autograd.grad(y, m.weight, gy, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * 9 * C * O
    print("%4d -> %4d @%dx%d  wgrad %.3f ms  %.1f TF/s  |gw| %.6e" % (C, O, H, W, ms, fl / ms / 1e9, float(gw.double().norm())))
    del m, x, y, gy, gw
