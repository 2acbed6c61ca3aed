"""Oracle (test infrastructure): the joint regression -> rasteriser -> projector composition on CPU.

The reference has no joint training; the product defines one in ``emlight_amd/joint.py``.  This file
restates that composition from the reference's own pieces so the HIP step can be checked end to end:
the guide map follows ``GenProjector/data.py:86-102`` (per-anchor light = distribution x intensity x rgb_ratio,
lobes of width .0025 on the Fibonacci anchors through ``convert_to_panorama``, plus the ambient term) with the
regression network's output scaling (``RegressionNetwork/data.py:70-73``: intensity x alpha / 500, ambient x alpha /
(128*256)) folded in: lobe amplitude = intensity_pred x 500 x 0.01, ambient_pred added as is.
"""
import contextlib

import torch
import torch.nn.functional as F

from .densenet import regression_loss
from .rasteriser import convert_to_panorama
from .sinkhorn import sphere_points


def predicted_gaussian_map(pred, ln, height=128):
    B = pred["distribution"].shape[0]
    dist_ = pred["distribution"].view(B, ln, 1).repeat(1, 1, 3)
    inten = (pred["intensity"].view(B, 1, 1) * 500.0 * 0.01).repeat(1, ln, 3)
    rgb = pred["rgb_ratio"].view(B, 1, 3).repeat(1, ln, 1)
    dirs = torch.from_numpy(sphere_points(ln)).float().view(1, ln * 3).repeat(B, 1)
    size = torch.ones((B, ln)) * 0.0025
    env = convert_to_panorama(dirs, size, (dist_ * inten * rgb).view(B, ln * 3), height=height)
    return env + pred["ambient"].view(B, 3, 1, 1)


def joint_generator_step(encoder, pix2pix, opt_E, opt_G, batch, emd_fn, ln):
    """Encoder + generator half of ``emlight_amd/joint.py::JointTrainer.generator_step``: regression loss + generator
    losses on the predicted guide map -> one backward -> Adam on encoder and generator.  ``pix2pix`` is a Pix2PixModel
    whose SphereNet ops run on stock ops (``oracle.stock_sphere_ops()``).  Gradients are left in ``.grad``."""
    pred = encoder(batch["crop"])
    l_reg, terms = regression_loss(pred, batch, emd_fn, ln)
    gmap = predicted_gaussian_map(pred, ln)
    crop128 = F.interpolate(batch["crop"], size=(128, 128), mode="bilinear", align_corners=False)
    data = {"input": gmap, "crop": crop128, "warped": batch["warped"], "map": batch["map"]}
    g_losses, fake = pix2pix(data, mode="generator")
    opt_E.zero_grad(set_to_none=True)
    opt_G.zero_grad()
    (l_reg + sum(g_losses.values()).mean()).backward()
    opt_E.step()
    opt_G.step()
    return {"pred": pred, "terms": terms, "g_losses": g_losses, "gmap": gmap, "fake": fake, "data": data}


def joint_discriminator_step(pix2pix, opt_D, data):
    """Discriminator half (``JointTrainer.discriminator_step``): losses on the detached map with the generator as the
    first half left it (already updated, as in ``GenProjector/train.py:33-37``) -> backward -> Adam on D."""
    opt_D.zero_grad()
    d_losses = pix2pix(dict(data, input=data["input"].detach()), mode="discriminator")
    sum(d_losses.values()).mean().backward()
    opt_D.step()
    return d_losses


def joint_step(encoder, pix2pix, opt_E, opt_G, opt_D, batch, emd_fn, ln):
    """One joint iteration, same order of operations as ``emlight_amd/joint.py::JointTrainer.step``."""
    out = joint_generator_step(encoder, pix2pix, opt_E, opt_G, batch, emd_fn, ln)
    out["d_losses"] = joint_discriminator_step(pix2pix, opt_D, out["data"])
    return out


@contextlib.contextmanager
def stock_rasteriser():
    """Inside the block the product's ``convert_to_panorama`` call sites run the oracle's CPU rasteriser (for the
    CPU-only distributed tests of the joint trainer; the product's own is HIP-only)."""
    from emlight_amd.GenProjector import data as pdata
    from emlight_amd.RegressionNetwork import util as rutil

    def cpu_rasteriser(dirs, sizes, colors, pano_hw=(128, 256)):
        return convert_to_panorama(dirs, sizes, colors, height=pano_hw[0])
    saved = pdata.convert_to_panorama, rutil.convert_to_panorama
    pdata.convert_to_panorama = rutil.convert_to_panorama = cpu_rasteriser
    try:
        yield
    finally:
        pdata.convert_to_panorama, rutil.convert_to_panorama = saved
