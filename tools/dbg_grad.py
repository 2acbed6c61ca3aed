import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from emlight_amd.RegressionNetwork.DenseNet import DenseNet
crop_hw, B = (64, 96), 2
ref = oracle.OracleDenseNet(anchors=32, crop_hw=crop_hw)
sd = oracle.deterministic_state_dict(ref.state_dict(), seed=5)
ref.load_state_dict(sd)
net = DenseNet(anchors=32, crop_hw=crop_hw, engine="hip").cuda(); net.load_state_dict(sd)
ref.train(); net.train()
g = np.random.default_rng([4, B])
x = torch.from_numpy(g.random((B, 3) + crop_hw, dtype=np.float32))
KEYS = ("distribution", "intensity", "rgb_ratio", "ambient")
w = {k: torch.from_numpy(g.standard_normal(s).astype(np.float32)) for k, s in (("distribution", (B, 32)), ("intensity", (B, 1)), ("rgb_ratio", (B, 3)), ("ambient", (B, 3)))}
po = ref(x); sum((po[k] * w[k]).sum() for k in KEYS).backward()
pg = net(x.cuda()); sum((pg[k] * w[k].cuda()).sum() for k in KEYS).backward()
nr, ng = dict(ref.named_parameters()), dict(net.named_parameters())
rows = []
for name, pr in nr.items():
    a, b = pr.grad.numpy(), ng[name].grad.cpu().numpy()
    rows.append((float(np.abs(a - b).max() / (np.abs(a).max() + 1e-12)), name, float(np.abs(a).max())))
for e, n, m in rows:
    if "features" in n and (e > 2e-3 or n.endswith("conv0.weight") or "last_norm" in n or "transition" in n):
        print("%-55s relerr %.3e  max|g| %.3e" % (n, e, m))
bad = [r for r in rows if r[0] > 2e-3]
print("bad %d / %d" % (len(bad), len(rows)))
kinds = {}
for e, n, m in bad:
    k = n.split(".")[-2] + "." + n.split(".")[-1]
    kinds[k] = kinds.get(k, 0) + 1
print(kinds)

# ---- recompute transition3 backward with torch from the engine's own buffers
enc = net._hip
ws = list(enc._ws.values())[0]
blk = ws.blocks[2]; tr = blk["trans"]
f = net.features
X3 = blk["X"][:, :342].double()
sc, sh = tr["scale"][:342].double(), tr["shift"][:342].double()
a = torch.relu(X3 * sc + sh)
Bn, Hb, Wb = 2, blk["H"], blk["W"]
ap = a.view(Bn, Hb // 2, 2, Wb // 2, 2, 342).mean(dim=(2, 4)).reshape(-1, 342)
Wt = f.transition3.conv.weight.detach().double().view(171, 342)
T2 = ap @ Wt.t()
print("T recompute err", float((T2 - tr["T"][:, :171].double()).abs().max()))
GF = ws.bwd.GF[:, :171].double()
Traw = tr["T"][:, :171].double()
mu, istd = tr["tmean"].double(), tr["tistd"].double()
gam = f.last_norm3.weight.detach().double()
that = (Traw - mu) * istd
n = Traw.shape[0]
dT = gam * istd * (GF - GF.mean(0) - that * (GF * that).mean(0))
dap = dT @ Wt
da = dap.view(Bn, Hb // 2, 1, Wb // 2, 1, 342).expand(Bn, Hb // 2, 2, Wb // 2, 2, 342).reshape(-1, 342) * 0.25
dam = torch.where(X3 * sc + sh > 0, da, torch.zeros_like(da))
S1 = dam.sum(0)
print("torch-recomputed dbeta_t vs engine:", float((S1 - f.transition3.norm.bias.grad.double()).abs().max()),
      " vs oracle:", float((S1.cpu() - ref.features.transition3.norm.bias.grad.double()).abs().max()), "scale", float(S1.abs().max()))
gT = ref.features.transition3.conv.weight.grad
print("blk mean check", float((blk["mean"][:342].double() - blk["X"][:, :342].double().mean(0)).abs().max()))
d = (S1.cpu() - ref.features.transition3.norm.bias.grad.double()).abs().numpy()
np.set_printoptions(precision=2, linewidth=200)
print("per-channel |err| x1e3 (342 ch in rows of 12; first 150 = block input):")
print((d * 1e3)[:150].reshape(-1, 15))
print((d * 1e3)[150:].reshape(-1, 12))
