"""Projector inputs.  The reference's ``LavalIndoorDataset.__getitem__`` (``GenProjector/data.py:58-108``)
rasterises the ground-truth Gaussian map PER SAMPLE on the GPU inside the loader; here the whole batch is one
call of the HIP rasteriser (``eml_sg_rasterise_f32``) in the training step.  EXR I/O is out of scope; the
synthetic generator follows SURVEY 8d."""
import torch
import torch.nn.functional as F

from ..RegressionNetwork.data import synthetic_batch
from ..RegressionNetwork.util import convert_to_panorama, sphere_points


_ANCHORS = {}


def anchor_dirs(ln, dev):
    """The Fibonacci anchors as a (1, 3 ln) device tensor, built once per (ln, device): the joint step calls ``gaussian_map`` every
    iteration, and a fresh ``torch.from_numpy(...).to(dev)`` there is a pageable host-to-device copy -- a host synchronisation
    in the middle of the step (round 6: every such copy on the iteration's path is gone, ``tools/capture_probe.py``)."""
    key = (int(ln), str(dev))
    if key not in _ANCHORS:
        _ANCHORS[key] = torch.from_numpy(sphere_points(ln)).float().view(1, ln * 3).to(dev)
    return _ANCHORS[key]


def gaussian_map(distribution, intensity, rgb_ratio, ambient, alpha=None, ln=128, pano_hw=(128, 256)):
    """``data.py:86-102``: light = dist * (intensity*0.01) * rgb per anchor, SG lobes of width .0025 on the
    Fibonacci anchors, + ambient / (H*W), * alpha.  All arguments are batched device tensors."""
    B = distribution.shape[0]
    dev = distribution.device
    dirs = anchor_dirs(ln, dev).expand(B, -1).contiguous()
    size = torch.full((B, ln), 0.0025, device=dev)
    light = (distribution.view(B, ln, 1) * (intensity.view(B, 1, 1) * 0.01) * rgb_ratio.view(B, 1, 3))
    env = convert_to_panorama(dirs, size, light.reshape(B, ln * 3).contiguous(), pano_hw=pano_hw)
    env = env + (ambient / (pano_hw[0] * pano_hw[1])).view(B, 3, 1, 1)
    return env if alpha is None else env * alpha.view(B, 1, 1, 1)


def projector_batch(batch, device, ln=128, pano_hw=(128, 256), seed=1234):
    """Synthetic ``{'input','crop','warped','map'}`` batch on ``device``."""
    p = synthetic_batch(batch, ln, (128, 128), seed=seed, device=device)
    g = torch.Generator().manual_seed(seed + 7)
    inp = gaussian_map(p["distribution"], p["intensity"] * 500.0, p["rgb_ratio"], p["ambient"] * pano_hw[0] * pano_hw[1],
                       ln=ln, pano_hw=pano_hw)
    noise = F.interpolate(torch.empty(batch, 1, 8, 16).uniform_(0.5, 1.5, generator=g), size=pano_hw,
                          mode="bilinear", align_corners=False).to(device)
    warped = inp * noise
    luma = 0.3 * warped[:, 0] + 0.59 * warped[:, 1] + 0.11 * warped[:, 2]
    mask = (luma > 0.05 * luma.amax(dim=(1, 2), keepdim=True)).float().unsqueeze(1)
    return {"input": inp, "crop": p["crop"], "warped": warped, "map": mask}
