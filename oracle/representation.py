"""Oracle (test infrastructure): EMLight's ground-truth parametrisation of an HDR panorama.

CPU float64 numpy restatement of ``extract_mesh`` (``RegressionNetwork/representation/distribution_representation.py:65-120``):
nearest-anchor Voronoi binning of a steradian-weighted panorama into (distribution, intensity, rgb_ratio, ambient) -- the
inverse of the spherical-Gaussian rasteriser and the producer of the pickles ``data.py`` reads.
"""
import numpy as np

from .sinkhorn import sphere_points


class ExtractMesh:
    def __init__(self, h=128, w=256, ln=64):
        self.h, self.w, self.ln = h, w, ln
        # :68-72 -- sin of the pixel-centre polar angle, constant along a row
        self.steradian = np.sin((np.arange(h) + 0.5) / h * np.pi)[:, None, None] * np.ones((1, w, 1))
        # :74-83 -- view vectors on the ENDPOINT-INCLUSIVE grid linspace(0, pi, h) x linspace(0, 2 pi, w)
        theta = np.linspace(0, np.pi, num=h)[:, None]
        phi = np.linspace(0, 2 * np.pi, num=w)[None, :]
        xyz = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta) * np.ones_like(phi)], -1)
        self.anchors = sphere_points(ln)
        # :86-87 -- nearest anchor of every pixel
        d = np.linalg.norm(xyz[:, :, None, :] - self.anchors[None, None], axis=-1)
        self.idx = np.argmin(d, axis=-1)

    def compute(self, hdr):
        """hdr (h, w, 3) -> ({'distribution' (ln,), 'intensity' (), 'rgb_ratio' (3,), 'ambient' (3,)}, map (h, w, 1))."""
        hdr = self.steradian * hdr                                                   # :92
        lum = 0.3 * hdr[..., 0] + 0.59 * hdr[..., 1] + 0.11 * hdr[..., 2]            # :93
        mask = (lum > lum.max() * 0.05)[..., None]                                   # :94-97
        light, remain = hdr * mask, hdr * (1 - mask)
        ambient = remain.sum(axis=(0, 1))                                            # :101
        anchors = np.zeros((self.ln, 3))
        np.add.at(anchors, self.idx.reshape(-1), light.reshape(-1, 3))               # :104-107
        energy = 0.3 * anchors[..., 0] + 0.59 * anchors[..., 1] + 0.11 * anchors[..., 2]
        rgb = anchors.sum(0)
        intensity = np.linalg.norm(rgb)
        return ({"distribution": energy / energy.sum(), "intensity": intensity, "rgb_ratio": rgb / intensity,
                 "ambient": ambient}, mask)
