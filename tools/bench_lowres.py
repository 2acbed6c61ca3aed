"""Round 6: the low-resolution wide SphereConv layers of the ngf = 64 projector (B = 32 per GPU; the discriminators see 64) --
the footprint gather-GEMM (csrc/gather_gemm3.h, EML_LOWRES=force) against round 5's dispatch (EML_LOWRES=off: gather_gemm2 or
im2col + the library GEMM on its recorded selection + col2im): forward, input gradient, weight gradient, each timed alone with
HIP events.  One JSON line per layer.     python tools/bench_lowres.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _runtime  # noqa: E402
_runtime.entry_point_defaults()
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [  # name, C, O, H, W, batch multiplier
    ("head_0 conv 1024->1024 @4x8", 1024, 1024, 4, 8, 1),
    ("head_0 gamma|beta 128->2048 @4x8", 128, 2048, 4, 8, 1),
    ("G_middle conv 1024->1024 @8x16", 1024, 1024, 8, 16, 1),
    ("G_middle gamma|beta 128->2048 @8x16", 128, 2048, 8, 16, 1),
    ("up_0 conv_0 1024->512 @16x32", 1024, 512, 16, 32, 1),
    ("up_0 conv_1 512->512 @16x32", 512, 512, 16, 32, 1),
    ("up_0 gamma|beta 128->2048 @16x32", 128, 2048, 16, 32, 1),
    ("up_0 gamma|beta 128->1024 @16x32", 128, 1024, 16, 32, 1),
    ("up_1 conv_0 512->256 @32x64", 512, 256, 32, 64, 1),
    ("up_1 conv_1 256->256 @32x64", 256, 256, 32, 64, 1),
    ("up_1 gamma|beta 128->1024 @32x64", 128, 1024, 32, 64, 1),
    ("up_1 gamma|beta 128->512 @32x64", 128, 512, 32, 64, 1),
    ("D 256->512 @16x32 (2B)", 256, 512, 16, 32, 2),
    ("D 256->512 @8x16 (2B)", 256, 512, 8, 16, 2),
]


def events(fn, reps=10):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    only = os.environ.get("BENCH_ONLY")
    for name, C, O, H, W, mult in LAYERS:
        if only and only not in name:
            continue
        bb = B * mult
        m = SphereConv2D(C, O, stride=1).cuda()
        x = torch.randn(bb, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        gflop = 2.0 * bb * H * W * 9 * C * O / 1e9
        row = {"layer": name, "B": bb, "gflop": round(gflop, 1)}
        for mode in ("off", "force"):
            SphereConv2D.lowres = mode
            with torch.no_grad():
                t_f = events(lambda: m(x))
            xg = x.clone().requires_grad_(True)
            y = m(xg)
            gy = torch.randn_like(y)
            t_d = events(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
            t_w = events(lambda: torch.autograd.grad(y, m.weight, gy, retain_graph=True))
            row[mode] = {"fwd_ms": round(t_f, 4), "fwd_tflops": round(gflop / t_f, 1), "dgrad_ms": round(t_d, 4),
                         "dgrad_tflops": round(gflop / t_d, 1), "wgrad_ms": round(t_w, 4), "wgrad_tflops": round(gflop / t_w, 1)}
            del y, gy, xg
        print(json.dumps(row), flush=True)
        del m, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
