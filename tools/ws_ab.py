"""A/B of the two fused SphereConv forward kernels (interleaved vs wave-specialised, env EML_SPHERE_WS) per layer shape:
bitwise equality of forward output and input gradient, and their timings.    python tools/ws_ab.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [(128, 256, 128, 256), (128, 128, 128, 256), (128, 64, 128, 256), (64, 64, 128, 256), (128, 512, 64, 128),
          (256, 128, 64, 128), (128, 128, 64, 128), (512, 256, 32, 64), (128, 128, 16, 32)]
SphereConv2D.fused_min_bytes = 0


def events(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for C, O, H, W in LAYERS:
    m = SphereConv2D(C, O).cuda()
    m.weight.requires_grad_(False)
    m.bias.requires_grad_(False)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    gflop = 2.0 * B * H * W * 9 * C * O / 1e9
    res = {}
    for ws in ("0", "1"):
        os.environ["EML_SPHERE_WS"] = ws
        with torch.no_grad():
            y = m(x)
            t_f = events(lambda: m(x))
        xg = x.clone().requires_grad_(True)
        yg = m(xg)
        gy = torch.ones_like(yg) * 0.5 + yg.detach() * 0.1
        gx, = torch.autograd.grad(yg, xg, gy, retain_graph=True)
        t_d = events(lambda: torch.autograd.grad(yg, xg, gy, retain_graph=True))
        res[ws] = (y, gx, t_f, t_d)
    eq_f = torch.equal(res["0"][0], res["1"][0])
    eq_d = torch.equal(res["0"][1], res["1"][1])
    print("C=%4d O=%4d %3dx%3d  fwd %.3f -> %.3f ms (%.1f -> %.1f TF/s)  dgrad %.3f -> %.3f ms   equal: %s %s" % (
        C, O, H, W, res["0"][2], res["1"][2], gflop / res["0"][2], gflop / res["1"][2], res["0"][3], res["1"][3], eq_f, eq_d),
        flush=True)
