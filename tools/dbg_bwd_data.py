import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emlight_amd import _lib
L, p = _lib.lib(), _lib.ptr
st = _lib.current_stream()
torch.manual_seed(0)
dev = "cuda"
def run(pool, B, Hin, Win, Cin, Cout):
    Kp = (Cin + 15) // 16 * 16; Ko = (Cout + 15) // 16 * 16
    ldx = Kp + 16
    Pin = B * Hin * Win
    P = Pin // 4 if pool else Pin
    X = torch.randn(Pin, ldx, device=dev)
    s1 = torch.zeros(Kp, device=dev); t1 = torch.zeros(Kp, device=dev)
    s1[:Cin] = torch.rand(Cin, device=dev) + 0.5; t1[:Cin] = torch.randn(Cin, device=dev) * 0.3
    mean = torch.randn(ldx, device=dev) * 0.1; istd = torch.rand(ldx, device=dev) + 0.5
    ld_dy = Ko + 16
    DY = torch.randn(P, ld_dy, device=dev); Zr = torch.randn(P, Ko, device=dev)
    cA = torch.zeros(Ko, device=dev); cB = torch.zeros(Ko, device=dev); cC = torch.zeros(Ko, device=dev)
    cA[:Cout] = torch.randn(Cout, device=dev); cB[:Cout] = torch.randn(Cout, device=dev) * 0.1; cC[:Cout] = torch.randn(Cout, device=dev) * 0.1
    W = torch.randn(Cout, Cin, device=dev) / np.sqrt(Cout)
    Wd = torch.empty(Kp * Ko, device=dev)
    _lib.check(L.eml_dense_permute_w1_bwd_f32(p(W), Cout, Cin, Kp, Ko, p(Wd), st), "perm")
    DA = torch.full((Pin, Kp), 7.0, device=dev)
    G = 512
    part = torch.zeros(G * Kp * 2, dtype=torch.float64, device=dev)
    _lib.check(L.eml_dense_conv1x1_bwd_data_f32(p(DY), ld_dy, p(Zr), Ko, p(cA), p(cB), p(cC), Ko, p(Wd), p(X), ldx, p(s1), p(t1), p(mean), p(istd), P, Hin, Win, int(pool), Kp, p(DA), p(part), G, st), "data")
    torch.cuda.synchronize()
    # reference
    dz = cA[:Cout] * DY[:, :Cout] + cB[:Cout] * Zr[:, :Cout] + cC[:Cout]
    da = dz.double() @ W.double()   # [P][Cin]
    pre = X[:, :Cin] * s1[:Cin] + t1[:Cin]
    if pool:
        da = da.view(B, Hin // 2, 1, Win // 2, 1, Cin).expand(B, Hin // 2, 2, Win // 2, 2, Cin).reshape(Pin, Cin) * 0.25
    dam = torch.where(pre > 0, da, torch.zeros_like(da))
    xh = (X[:, :Cin].double() - mean[:Cin].double()) * istd[:Cin].double()
    S1 = dam.sum(0); S2 = (dam * xh).sum(0)
    got = part.view(G, Kp, 2).sum(0)
    print("pool", pool, "Cin", Cin, "Cout", Cout, "DA err %.3e" % float((DA[:, :Cin].double() - dam).abs().max()), "pad DA max %.3e" % float(DA[:, Cin:].abs().max() if Kp > Cin else 0),
          "S1 err %.3e (scale %.2e)" % (float((got[:Cin, 0] - S1).abs().max()), float(S1.abs().max())), "S2 err %.3e (scale %.2e)" % (float((got[:Cin, 1] - S2).abs().max()), float(S2.abs().max())))
run(False, 2, 16, 24, 150 + 12 * 15, 48)
run(False, 2, 16, 24, 24, 48)
run(True, 2, 16, 24, 342, 171)
run(True, 2, 32, 48, 300, 150)
run(True, 3, 20, 28, 216, 108)
