"""How far are the two conv3x3-forward kernels from the reference's golden outputs (192x256, 96 anchors, O(100) logits)
and from the f64 truth of the same network?  EML_C3_TP=off / auto in one process each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from emlight_amd.RegressionNetwork.DenseNet import DenseNet
KEYS = ("distribution", "intensity", "rgb_ratio", "ambient")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "densenet_reference.npz")) if os.path.exists(
    os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "densenet_reference.npz")) else None
ref = oracle.OracleDenseNet(anchors=96, crop_hw=(192, 256))
sd = oracle.deterministic_state_dict(ref.state_dict(), seed=0)
ref.load_state_dict(sd)
net = DenseNet(anchors=96, crop_hw=(192, 256)).cuda(); net.load_state_dict(sd)
x = torch.from_numpy(np.random.default_rng([0]).random((2, 3, 192, 256), dtype=np.float32))
ref64 = oracle.OracleDenseNet(anchors=96, crop_hw=(192, 256)).double(); ref64.load_state_dict({k: v.double() for k, v in sd.items()})
for mode in ("eval", "train"):
    ref.train(mode == "train"); ref64.train(mode == "train"); net.train(mode == "train")
    with torch.no_grad():
        got = net(x.cuda()); w32 = ref(x); w64 = ref64.cuda()(x.double().cuda()) if False else ref64(x.double())
    for k in KEYS:
        a = got[k].cpu().double(); b = w32[k].double(); c = w64[k]
        print("%s %-12s |hip - cpu32| %.3e   |hip - f64| %.3e   |cpu32 - f64| %.3e   (max |value| %.1f)" % (
            mode, k, (a - b).abs().max(), (a - c).abs().max(), (b - c).abs().max(), c.abs().max()), flush=True)
