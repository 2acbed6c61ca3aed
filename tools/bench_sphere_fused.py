"""A few launches of the fused SphereConv kernels on one layer shape (for rocprofv3 --pmc runs).
    python tools/bench_sphere_fused.py [C O H W B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402

C, O, H, W, B = (int(a) for a in (sys.argv[1:6] + ["128", "256", "128", "256", "32"][len(sys.argv) - 1:]))
SphereConv2D.fused_min_bytes = 0
m = SphereConv2D(C, O).cuda()
x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
for _ in range(3):
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
torch.cuda.synchronize()
