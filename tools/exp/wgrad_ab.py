"""Weight gradient of SphereConv2D at the projector's layer shapes: time per call of the fused kernel (EML_WG_V1=1: the round-2
kernel) and the largest relative deviation from the unfused im2col + library GEMM.   python tools/exp/wgrad_ab.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emlight_amd.GenProjector.spherenet import SphereConv2D  # noqa: E402
from tools.sphere_layers import LAYERS, events  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
tot = 0.0
for name, C, O, H, W, stride in LAYERS:
    if C % 64:
        continue
    bb = B * 2 if name.startswith("D ") else B
    m = SphereConv2D(C, O, stride=stride).cuda()
    x = torch.randn(bb, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    po = (H // stride) * (W // stride)
    gflop = 2.0 * bb * po * 9 * C * O / 1e9
    res = {}
    for mode, thr in (("fused", 0), ("unfused", 1 << 62)):
        SphereConv2D.fused_min_bytes = thr
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        gy = torch.randn(y.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1)).contiguous(memory_format=torch.channels_last)
        res[mode] = torch.autograd.grad(y, m.weight, gy, retain_graph=True)[0]
        if mode == "fused":
            t_w = events(lambda: torch.autograd.grad(y, m.weight, gy, retain_graph=True), reps=10)
    err = float((res["fused"] - res["unfused"]).abs().max() / res["unfused"].abs().max())
    tot += t_w
    print(json.dumps({"layer": name, "wgrad_ms": round(t_w, 3), "tflops": round(gflop / t_w, 1), "rel_err": err,
                      "v1": os.environ.get("EML_WG_V1", "0")}), flush=True)
print(json.dumps({"total_ms": round(tot, 3), "v1": os.environ.get("EML_WG_V1", "0")}))
