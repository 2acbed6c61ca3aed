"""Oracle (test infrastructure): spherical-Gaussian lobes -> equirectangular panorama.

CPU f32 restatement of ``convert_to_panorama`` -- ``RegressionNetwork/util.py:222-245``
(identical copies: ``representation/util.py:205-228``, ``GenProjector/util.py:346-369``;
variable-latitude form ``RegressionNetwork/panorama.py:68-82,142-152``).  The
panorama height is a parameter (reference: 128) and nothing calls ``.cuda()``.
"""
import numpy as np
import torch


def pano_grid(height=128):
    """Unit view vectors of the ``height x 2*height`` equirect grid, ``(3, H, 2H)`` f32.

    theta = (h + .5) pi / H, phi = (w + .5) pi / H; xyz = (sin t cos p, sin t sin p, cos t).
    Follows ``util.py:223-233`` (f32 torch arithmetic).
    """
    lat, lon = torch.meshgrid(
        [torch.arange(height, dtype=torch.float),
         torch.arange(2 * height, dtype=torch.float)], indexing="ij")
    lat = lat.add(0.5).mul(np.pi / height)
    lon = lon.add(0.5).mul(np.pi / height)
    return torch.stack((torch.sin(lat) * torch.cos(lon),
                        torch.sin(lat) * torch.sin(lon),
                        torch.cos(lat)))


def convert_to_panorama(dirs, sizes, colors, height=128):
    """``lights[b,c,h,w] = sum_i colors[b,3i+c] * exp((dirs[b,3i:3i+3].xyz[:,h,w] - 1)/sizes[b,i])``.

    dirs ``(B, 3N)``, sizes ``(B, N)``, colors ``(B, 3N)`` -> ``(B, 3, H, 2H)``.
    Follows the per-light accumulation loop at ``util.py:235-244`` (same op order).
    """
    xyz = pano_grid(height).to(dirs.dtype)
    B = colors.shape[0]
    n = colors.shape[1] // 3
    out = torch.zeros((B, 3, height, 2 * height), dtype=dirs.dtype)
    flat = xyz.view(3, -1)
    for i in range(n):
        dot = torch.matmul(dirs[:, 3 * i:3 * i + 3], flat).view(-1, height, 2 * height)
        lobe = torch.exp((dot - 1) / sizes[:, i].view(-1, 1, 1))
        out = out + colors[:, 3 * i:3 * i + 3][:, :, None, None] * lobe[:, None, :, :]
    return out
