"""GPU: one projector iteration (G step + D step) with the ground-truth Gaussian map produced by the HIP
rasteriser, and the batched GT map against the oracle's per-sample recipe (GenProjector/data.py:86-102)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def test_gaussian_map_batched_vs_oracle():
    from emlight_amd.GenProjector.data import gaussian_map
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    B, ln = 3, 128
    p = synthetic_batch(B, ln, (32, 32), seed=5)
    inten, amb = p["intensity"] * 300.0, p["ambient"] * 128 * 256
    got = gaussian_map(p["distribution"].cuda(), inten.cuda(), p["rgb_ratio"].cuda(), amb.cuda(), ln=ln).cpu()
    dirs = torch.from_numpy(oracle.sphere_points(ln)).float().view(1, ln * 3)
    size = torch.full((1, ln), 0.0025)
    for b in range(B):  # the reference's per-item code path
        light = (p["distribution"][b].view(1, ln, 1).repeat(1, 1, 3) * (inten[b] * 0.01).view(1, 1, 1).repeat(1, ln, 3)
                 * p["rgb_ratio"][b].view(1, 1, 3).repeat(1, ln, 1)).view(1, ln * 3)
        want = oracle.convert_to_panorama(dirs, size, light).view(3, 128, 256) + (amb[b] / (128 * 256)).view(3, 1, 1)
        np.testing.assert_allclose(got[b].numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * float(want.max()))


def test_projector_iteration_on_gpu():
    from emlight_amd.GenProjector import data, networks
    from emlight_amd.GenProjector.model_trainer import Trainer
    torch.manual_seed(0)
    tr = Trainer(networks.default_options(ngf=4, ndf=4), device="cuda")
    batch = data.projector_batch(2, "cuda", seed=3)
    assert batch["input"].shape == (2, 3, 128, 256) and batch["map"].shape == (2, 1, 128, 256)
    tr.step(batch)
    losses = tr.get_latest_losses()
    assert set(losses) == {"GAN", "GAN_Feat", "COS", "D_Fake", "D_real"}
    assert all(bool(torch.isfinite(v).all()) for v in losses.values())
    assert tr.generated.shape == (2, 3, 128, 256)
