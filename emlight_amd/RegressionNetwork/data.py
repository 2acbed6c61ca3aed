"""Datasets for the regression network.

The reference's ``ParameterDataset`` (``RegressionNetwork/data.py:20-87``) reads the
licence-restricted Laval Indoor EXR crops + pickled GT parameters; that I/O is out of scope
(SURVEY C7).  ``SyntheticParameterDataset`` yields the same dict -- keys ``crop``,
``distribution``, ``intensity``, ``rgb_ratio``, ``ambient``, ``name`` with the same shapes
and value ranges -- from a seeded generator, and ``PickleParameterDataset`` reads the
reference's on-disk format (``representation/distribution_representation.py:116-119``)
when a directory of ``*.pickle`` + ``*.npy`` crops is supplied.
"""
import glob
import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset


def synthetic_batch(batch, anchors=128, crop_hw=(240, 320), seed=1234, device="cpu"):
    """One seeded synthetic batch: uniform crops, sparse-simplex distribution
    (mimics the >5%-of-max light mask of ``distribution_representation.py:96,110``),
    intensity U(.05,2), unit-L2 rgb ratio, ambient U(0,.3)."""
    g = torch.Generator().manual_seed(seed)
    crop = torch.rand(batch, 3, crop_hw[0], crop_hw[1], generator=g)
    d = torch.softmax(4.0 * torch.randn(batch, anchors, generator=g), dim=1)
    thr = torch.quantile(d, 0.75, dim=1, keepdim=True)
    d = torch.where(d < thr, torch.zeros_like(d), d)
    d = d / d.sum(1, keepdim=True)
    intensity = torch.empty(batch, 1).uniform_(0.05, 2.0, generator=g)
    rgb = torch.empty(batch, 3).uniform_(0.4, 0.7, generator=g)
    rgb = rgb / rgb.norm(dim=1, keepdim=True)
    ambient = torch.empty(batch, 3).uniform_(0.0, 0.3, generator=g)
    out = {"crop": crop, "distribution": d, "intensity": intensity, "rgb_ratio": rgb, "ambient": ambient}
    return {k: v.to(device) for k, v in out.items()}


class SyntheticParameterDataset(Dataset):
    def __init__(self, length=4096, anchors=128, crop_hw=(240, 320), seed=1234):
        self.length, self.anchors, self.crop_hw, self.seed = length, anchors, tuple(crop_hw), seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        b = synthetic_batch(1, self.anchors, self.crop_hw, seed=self.seed + idx)
        item = {k: v[0] for k, v in b.items()}
        item["name"] = "synthetic_%06d" % idx
        return item


class PickleParameterDataset(Dataset):
    """``<dir>/<name>.pickle`` = {'distribution','intensity','rgb_ratio','ambient'} (the
    reference's GT format) next to ``<name>.npy`` = tonemapped crop (3,H,W) f32 in [0,1].
    Scaling follows ``data.py:70-73``: intensity*alpha/500, ambient*alpha/(128*256) with the
    tonemap alpha stored in the pickle as 'alpha' (1.0 if absent)."""

    def __init__(self, root):
        self.items = sorted(glob.glob(os.path.join(root, "*.pickle")))
        if not self.items:
            raise FileNotFoundError("no *.pickle under %s" % root)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        path = self.items[idx]
        with open(path, "rb") as f:
            p = pickle.load(f)
        alpha = float(p.get("alpha", 1.0))
        crop = np.load(path[:-len(".pickle")] + ".npy").astype(np.float32)
        return {"crop": torch.from_numpy(crop),
                "distribution": torch.as_tensor(p["distribution"], dtype=torch.float32),
                "intensity": torch.as_tensor(p["intensity"], dtype=torch.float32).reshape(1) * alpha / 500.0,
                "rgb_ratio": torch.as_tensor(p["rgb_ratio"], dtype=torch.float32),
                "ambient": torch.as_tensor(p["ambient"], dtype=torch.float32) * alpha / (128 * 256),
                "name": os.path.basename(path)[:-len(".pickle")]}
