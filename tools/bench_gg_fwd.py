"""Forward of the fused SphereConv gather-GEMM at a few of the projector's shapes, B = 32: ms and TF/s per call (HIP events).
    python tools/bench_gg_fwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib  # noqa: E402
if os.environ.get("EML_LIB_PATH"):
    _lib.LIB_PATH = os.environ["EML_LIB_PATH"]
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402

SphereConv2D.fused_min_bytes = 0
B = 32
for C, O, H, W in [(128, 128, 128, 256), (128, 64, 128, 256), (64, 64, 128, 256), (256, 128, 64, 128), (128, 256, 64, 128)]:
    m = SphereConv2D(C, O).cuda()
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            y = m(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%4d -> %4d @%dx%d  fwd %.3f ms  %.1f TF/s   checksum %.6e" % (C, O, H, W, ms, 2.0 * B * H * W * 9 * C * O / ms / 1e9,
                                                                       float(y.double().abs().mean())))
