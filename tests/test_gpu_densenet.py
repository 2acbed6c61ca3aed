"""GPU parity: HIP DenseNet-BC engine (through the C ABI) vs the oracle / the reference's golden vectors.

north_star gate: regressed (distribution, intensity, RGB, ambient) tensors within 1e-4 abs."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
OUT_ATOL = 1e-4
KEYS = ("distribution", "intensity", "rgb_ratio", "ambient")


def _pair(anchors, crop_hw, seed):
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    ref = oracle.OracleDenseNet(anchors=anchors, crop_hw=crop_hw)
    sd = oracle.deterministic_state_dict(ref.state_dict(), seed=seed)
    ref.load_state_dict(sd)
    net = DenseNet(anchors=anchors, crop_hw=crop_hw, engine="hip").cuda()
    net.load_state_dict(sd)
    return ref, net


@pytest.mark.parametrize("crop_hw,B", [((64, 96), 2), ((32, 32), 3), ((96, 160), 1)])
@pytest.mark.parametrize("train", [False, True])
def test_forward_small_vs_oracle(crop_hw, B, train):
    ref, net = _pair(32, crop_hw, seed=3)
    ref.train(train), net.train(train)
    x = torch.from_numpy(np.random.default_rng([1, B]).random((B, 3) + crop_hw, dtype=np.float32))
    with torch.no_grad():
        want, got = ref(x), net(x.cuda())
        fw = torch.nn.functional.avg_pool2d(torch.relu(ref.features_forward(x)), 4).reshape(B, -1)
        fg = net.pooled_features(x.cuda())
    np.testing.assert_allclose(fg.cpu().numpy(), fw.numpy(), rtol=3e-5, atol=2e-5)
    for k in KEYS:
        # synthetic eval-mode weights give outputs of O(100): 1e-4 abs + 1e-5 rel
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), rtol=1e-5, atol=OUT_ATOL)
    if train:  # running statistics follow nn.BatchNorm2d
        for name in ("features.norm0", "features.denseblock2.denselayer5.norm2", "features.last_norm3",
                     "features.transition1.norm"):
            a, b = dict(ref.named_modules())[name], dict(net.named_modules())[name]
            np.testing.assert_allclose(b.running_mean.cpu().numpy(), a.running_mean.numpy(), rtol=0, atol=1e-5)
            np.testing.assert_allclose(b.running_var.cpu().numpy(), a.running_var.numpy(), rtol=1e-4, atol=1e-6)
            assert int(b.num_batches_tracked) == 2  # two train-mode forward calls above (nn.BatchNorm2d counts calls)


def test_forward_golden_reference_geometry(golden_densenet):
    """192x256 crops, 96 anchors: outputs of the REAL reference DenseNet (eval and train mode)."""
    g = golden_densenet
    _, net = _pair(96, (192, 256), seed=0)
    x = torch.from_numpy(np.random.default_rng([0]).random((2, 3, 192, 256), dtype=np.float32)).cuda()
    with torch.no_grad():
        net.eval()
        pe = net(x)
        net.train()
        pt = net(x)
    for k in KEYS:
        np.testing.assert_allclose(pe[k].cpu().numpy(), g["eval/" + k], rtol=0, atol=OUT_ATOL)
        np.testing.assert_allclose(pt[k].cpu().numpy(), g["train/" + k], rtol=0, atol=OUT_ATOL)


def test_forward_cfg1_240x320_128_anchors(golden_densenet):
    g = golden_densenet
    _, net = _pair(128, (240, 320), seed=1)
    net.eval()
    x = torch.from_numpy(np.random.default_rng([2]).random((1, 3, 240, 320), dtype=np.float32)).cuda()
    with torch.no_grad():
        p = net(x)
    for k in KEYS:
        np.testing.assert_allclose(p[k].cpu().numpy(), g["cfg1/" + k], rtol=0, atol=OUT_ATOL)
