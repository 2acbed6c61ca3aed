"""SphereNet ops of the projector (reference ``models/networks/spherenet/sphere_cnn.py``).

``SphereConv2D`` = gather the 3x3 tangent-plane neighbourhood of every output pixel with
``grid_sample`` into a (3H', 3W') tensor, then a stride-3 3x3 convolution (``sphere_cnn.py:111-124``).
The sampling pattern (``cal_index``/``gen_grid_coordinates``, ``:31-84``) depends only on (H, W, stride):
it is built once, vectorised in float64 numpy (the reference loops H*W times in Python), and cached.
"""
from functools import lru_cache

import numpy as np
import torch
from torch import nn
from torch.nn.parameter import Parameter


@lru_cache(None)
def sphere_sampling_grid(h, w, stride=1):
    """``(1, 3*h/stride, 3*w/stride, 2)`` float32 grid for ``F.grid_sample`` (x, y in [-1, 1]).

    Row/col of tap (a, b) around pixel (r, c) follow the gnomonic projection of ``sphere_cnn.py:31-58``;
    the centre tap is the pixel itself; columns wrap modulo w; normalisation ``v*2/size - 1`` and the
    (y, x) -> (x, y) swap follow ``:75-84``.
    """
    rows = np.arange(0, h, stride, dtype=np.float64)
    cols = np.arange(0, w, stride, dtype=np.float64)
    phi = -((rows + 0.5) / h * np.pi - np.pi / 2)[:, None, None, None]          # (H',1,1,1)
    theta = ((cols + 0.5) / w * 2 * np.pi - np.pi)[None, :, None, None]           # (1,W',1,1)
    d_phi, d_theta = np.pi / h, 2 * np.pi / w
    tx, ty, sec = np.tan(d_theta), np.tan(d_phi), 1.0 / np.cos(d_theta)
    x = np.array([[-tx, 0.0, tx]] * 3)                                            # (3,3) tangent-plane x
    y = np.array([[sec * ty, ty, sec * ty], [0.0, 1.0, 0.0], [-sec * ty, -ty, -sec * ty]])
    x[1, 1], y[1, 1] = 1.0, 1.0                                                   # placeholder, overwritten below
    x, y = x[None, None], y[None, None]
    rho = np.sqrt(x * x + y * y)
    v = np.arctan(rho)
    new_phi = np.arcsin(np.cos(v) * np.sin(phi) + y * np.sin(v) * np.cos(phi) / rho)
    new_theta = theta + np.arctan(x * np.sin(v) / (rho * np.cos(phi) * np.cos(v) - y * np.sin(phi) * np.sin(v)))
    shape = (len(rows), len(cols), 3, 3)
    new_r = np.broadcast_to((-new_phi + np.pi / 2) * h / np.pi - 0.5, shape).copy()
    new_c = np.broadcast_to(((new_theta + np.pi) * w / 2 / np.pi - 0.5 + w) % w, shape).copy()
    new_r[:, :, 1, 1] = rows[:, None]
    new_c[:, :, 1, 1] = cols[None, :]
    gy = new_r * 2 / h - 1                                                        # (H',W',3,3)
    gx = new_c * 2 / w - 1
    grid = np.stack([gx, gy], axis=-1)                                            # (H',W',3,3,2)
    grid = grid.transpose(0, 2, 1, 3, 4).reshape(1, 3 * len(rows), 3 * len(cols), 2)
    return torch.from_numpy(np.ascontiguousarray(grid)).float()


class SphereConv2D(nn.Module):
    """3x3 spherical convolution, same parameters (``weight`` (out,in,3,3), ``bias``) and init as the
    reference (``sphere_cnn.py:87-109``)."""

    def __init__(self, in_c, out_c, stride=1, bias=True, mode="bilinear"):
        super().__init__()
        self.in_c, self.out_c, self.stride, self.mode = in_c, out_c, stride, mode
        self.weight = Parameter(torch.empty(out_c, in_c, 3, 3))
        if bias:
            self.bias = Parameter(torch.empty(out_c))
        else:
            self.register_parameter("bias", None)
        self._grids = {}
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=np.sqrt(5))
        if self.bias is not None:
            self.bias.data.zero_()

    def grid_for(self, x):
        key = (x.shape[2], x.shape[3], x.device)
        g = self._grids.get(key)
        if g is None:
            g = sphere_sampling_grid(x.shape[2], x.shape[3], self.stride).to(x.device)
            self._grids = {key: g}
        return g

    def forward(self, x):
        grid = self.grid_for(x).expand(x.shape[0], -1, -1, -1)
        # torch >= 1.3 default align_corners=False, zero padding: what the reference runs with today
        x = nn.functional.grid_sample(x, grid, mode=self.mode, align_corners=False)
        return nn.functional.conv2d(x, self.weight, self.bias, stride=3)
