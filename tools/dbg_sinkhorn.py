import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
z = np.load("tests/golden/sinkhorn.npz")
for case in ["n256_blur05", "n128_blur05"]:
    x = torch.from_numpy(z[case + "/x"]); y = torch.from_numpy(z[case + "/y"])
    B, n = x.shape
    crit = SamplesLoss(blur=.05, anchors=n)
    r = crit.forward_raw(x.cuda().view(B, n, 1), y.cuda().view(B, n, 1))
    d = r["duals"].cpu().numpy(); w = z[case + "/duals"]
    for k in range(4):
        diff = d[k] - w[k]
        print(case, "plane", k, "max|diff| %.3e mean diff %.3e std diff %.3e  want range [%.3f %.3f]" % (np.abs(diff).max(), diff.mean(), diff.std(), w[k].min(), w[k].max()))
    print("loss", r["loss"].cpu().numpy(), z[case + "/loss"])
    # oracle on CPU for the same case: where do the duals of yx/xy sit?
    lo, aux = oracle.samples_loss(x.view(B, n, 1), y.view(B, n, 1), oracle.anchor_cost_matrix(n), blur=.05, return_aux=True)
    for k in range(4):
        print("   oracle-vs-golden plane", k, float(np.abs(aux["duals"][k].numpy() - w[k]).max()))
