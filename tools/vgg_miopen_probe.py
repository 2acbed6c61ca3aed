"""VGG19's convolutions at B = 32: this repo's gather-GEMM (planar tap table) against MIOpen (F.conv2d, channels-last f32),
forward and input gradient -- is the library faster on the ordinary 3x3 convolutions of the perceptual term?
    python tools/vgg_miopen_probe.py [B]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
from emlight_amd.GenProjector.spherenet import planar_conv3x3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda"


def events(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = {"hip_fwd": 0.0, "lib_fwd": 0.0, "hip_dgrad": 0.0, "lib_dgrad": 0.0}
for cin, cout, h, w, n in ((64, 64, 128, 256, 1), (64, 128, 64, 128, 1), (128, 128, 64, 128, 1), (128, 256, 32, 64, 1), (256, 256, 32, 64, 3),
                           (256, 512, 16, 32, 1), (512, 512, 16, 32, 3), (512, 512, 8, 16, 1)):
    x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wgt = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    wcl = wgt.contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=dev)
    gflop = 2.0 * B * h * w * 9 * cin * cout / 1e9
    gy = torch.randn(B, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    row = {"layer": "%d->%d @%dx%d x%d" % (cin, cout, h, w, n), "gflop": round(gflop, 1)}
    with torch.no_grad():
        t = events(lambda: planar_conv3x3(x, wgt, bias, 1, 0.0))
        row["hip_fwd"] = [round(t, 3), round(gflop / t, 1)]
        t = events(lambda: F.relu_(F.conv2d(x, wcl, bias, padding=1)))
        row["lib_fwd"] = [round(t, 3), round(gflop / t, 1)]
    yh = planar_conv3x3(x, wgt, bias, 1, 0.0)
    t = events(lambda: torch.autograd.grad(yh, x, gy, retain_graph=True))
    row["hip_dgrad"] = [round(t, 3), round(gflop / t, 1)]
    yl = F.relu(F.conv2d(x, wcl, bias, padding=1))
    t = events(lambda: torch.autograd.grad(yl, x, gy, retain_graph=True))
    row["lib_dgrad"] = [round(t, 3), round(gflop / t, 1)]
    for k in tot:
        tot[k] += n * row[k][0]
    print(json.dumps(row), flush=True)
print(json.dumps({"sum_ms_over_the_stack (conv1_1 excluded)": {k: round(v, 2) for k, v in tot.items()},
                  "per_step": "2 forwards + 1 input gradient"}))
