"""How do the two conv0 kernels round?  Output error against F.conv2d in f64 (192 x 256, B = 2, the golden geometry): maximum,
RMS, and the MEAN SIGNED error in units of the output's RMS -- a biased rounding shows in the last.
    python tools/exp/conv0_rounding.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from emlight_amd import _lib  # noqa: E402

L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
torch.manual_seed(0)
B, H, W, G = 2, 192, 256, 512
x = torch.rand(B, 3, H, W, device="cuda")
w0 = torch.randn(24, 3, 3, 3, device="cuda") * 0.3
want = F.conv2d(x.double(), w0.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, 24)
stock = F.conv2d(x, w0, padding=1).permute(0, 2, 3, 1).reshape(-1, 24).double()
rms = float(want.pow(2).mean().sqrt())
rows = [("torch conv2d f32 (MIOpen)", stock)]
for name in ("eml_dense_conv0_fwd_f32", "eml_dense_conv0_fwd_mfma_f32"):
    Y = torch.zeros(B * H * W, 24, device="cuda")
    part = torch.zeros(G * 48, dtype=torch.float64, device="cuda")
    _lib.check(getattr(L, name)(p(x), p(w0), p(Y), 24, B, H, W, 24, p(part), G, st), name)
    rows.append((name, Y.double()))
for name, Y in rows:
    e = Y - want
    print("%-32s max |e| %.3e  rms e %.3e  mean e %+.3e  (in units of rms(out) = %.3f: %.2e / %.2e / %+.2e); mean e where out > 0: %+.2e, < 0: %+.2e"
          % (name, float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(e.mean()), rms, float(e.abs().max()) / rms,
             float(e.pow(2).mean().sqrt()) / rms, float(e.mean()) / rms, float(e[want > 0].mean()), float(e[want < 0].mean())))
