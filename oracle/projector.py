"""Oracle (test infrastructure): the projector's SphereNet ops on stock PyTorch ops.

CPU restatement of ``GenProjector/models/networks/spherenet/sphere_cnn.py``: the gnomonic 3x3 sampling
pattern (``cal_index`` ``:31-58``, ``gen_grid_coordinates`` ``:75-84``) and
``SphereConv2D.forward`` = ``grid_sample`` + stride-3 ``conv2d`` (``:111-124``); plus SPADE's modulation
formula (``normalization.py:113-115``) followed by the LeakyReLU of ``architecture.py:56-57``.

The product (``emlight_amd/GenProjector/spherenet.py``) has ONE execution path, the HIP kernels.
``stock_sphere_ops()`` lets a test swap the product's two dispatch points for these restatements, so that the
host logic (module wiring, losses, trainers, DDP) can be exercised on CPU and the HIP path can be compared
with the reference's own two ops on the GPU.  Only ``tests/`` may use it.
"""
import contextlib
from functools import lru_cache

import numpy as np
import torch
import torch.nn.functional as F


def tap_rows_cols(h, w, r):
    """Sampling rows/cols of the 3x3 taps for every column of image row ``r``: two ``(w, 3, 3)`` float64 arrays.

    ``cal_index`` (``sphere_cnn.py:31-58``) per pixel: the tangent-plane offsets ``(x, y)`` of ``get_xy``
    (``:10-28``) are projected back to the sphere around ``(phi, theta)``; the centre tap is the pixel itself;
    columns wrap modulo ``w``."""
    d_phi, d_theta = np.pi / h, 2 * np.pi / w
    tx, ty, sec = np.tan(d_theta), np.tan(d_phi), 1.0 / np.cos(d_theta)
    x = np.array([[-tx, 0.0, tx], [-tx, 1.0, tx], [-tx, 0.0, tx]])
    y = np.array([[sec * ty, ty, sec * ty], [0.0, 1.0, 0.0], [-sec * ty, -ty, -sec * ty]])
    phi = -((r + 0.5) / h * np.pi - np.pi / 2)
    rho = np.sqrt(x ** 2 + y ** 2)
    v = np.arctan(rho)
    new_phi = np.arcsin(np.cos(v) * np.sin(phi) + y * np.sin(v) * np.cos(phi) / rho)
    d_th = np.arctan(x * np.sin(v) / (rho * np.cos(phi) * np.cos(v) - y * np.sin(phi) * np.sin(v)))
    rows = np.empty((w, 3, 3))
    cols = np.empty((w, 3, 3))
    for c in range(w):
        theta = (c + 0.5) / w * 2 * np.pi - np.pi
        rows[c] = (-new_phi + np.pi / 2) * h / np.pi - 0.5
        cols[c] = ((theta + d_th + np.pi) * w / 2 / np.pi - 0.5 + w) % w
        rows[c, 1, 1], cols[c, 1, 1] = r, c
    return rows, cols


@lru_cache(None)
def sampling_grid(h, w, stride=1):
    """``gen_grid_coordinates`` (``sphere_cnn.py:75-84``): ``(1, 3h', 3w', 2)`` f32 grid, (x, y) in [-1, 1]."""
    rr = list(range(0, h, stride))
    cc = list(range(0, w, stride))
    grid = np.empty((len(rr), 3, len(cc), 3, 2))
    for i, r in enumerate(rr):
        rows, cols = tap_rows_cols(h, w, r)
        grid[i, :, :, :, 1] = (rows[cc] * 2 / h - 1).transpose(1, 0, 2)
        grid[i, :, :, :, 0] = (cols[cc] * 2 / w - 1).transpose(1, 0, 2)
    return torch.from_numpy(grid.reshape(1, 3 * len(rr), 3 * len(cc), 2)).float()


def sphere_conv(x, weight, bias, stride=1, residual=None, act_slope=1.0):
    """``SphereConv2D.forward`` (``sphere_cnn.py:111-124``): grid_sample (bilinear, torch >= 1.3 defaults:
    ``align_corners=False``, zero padding) then a stride-3 3x3 convolution.  ``residual`` / ``act_slope``: the product folds
    the ops that FOLLOW some convolutions into their kernel; here they are the reference's own separate ops --
    ``x_s + dx`` (``architecture.py:60``) and ``F.leaky_relu(., 2e-1)`` (``generator.py:84``) / ``nn.ReLU`` (``normalization.py:92``)."""
    grid = sampling_grid(x.shape[2], x.shape[3], stride).to(x.device).expand(x.shape[0], -1, -1, -1)
    y = F.conv2d(F.grid_sample(x, grid, mode="bilinear", align_corners=False), weight, bias, stride=3)
    if residual is not None:
        y = residual + y
    if act_slope != 1.0:
        y = F.relu(y) if act_slope == 0.0 else F.leaky_relu(y, act_slope)
    return y


def spade_modulate(normalized, actv, conv_gamma, conv_beta, slope=1.0):
    """``normalized * (1 + gamma) + beta`` (``normalization.py:108-115``), then ``leaky_relu(., slope)``."""
    gamma = sphere_conv(actv, conv_gamma.weight, conv_gamma.bias, conv_gamma.stride)
    beta = sphere_conv(actv, conv_beta.weight, conv_beta.bias, conv_beta.stride)
    out = normalized * (1 + gamma) + beta
    return out if slope == 1.0 else F.leaky_relu(out, slope)


def spade_norm_modulate(x, bn, actv, conv_gamma, conv_beta, slope=1.0, stats=None):
    """SPADE.forward (``normalization.py:101-115``): param-free norm, then the modulation, then the LeakyReLU of
    ``architecture.py:56-57``.  ``stats`` (a product-side optimisation) is ignored: the norm module does its own."""
    return spade_modulate(bn(x), actv, conv_gamma, conv_beta, slope)


# torchvision ``vgg19().features`` up to relu5_1: conv index -> (in, out)
_VGG19_CONVS = {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256), 14: (256, 256),
                16: (256, 256), 19: (256, 512), 21: (512, 512), 23: (512, 512), 25: (512, 512), 28: (512, 512)}
_VGG19_POOLS = (4, 9, 18, 27)
_VGG19_SLICE_ENDS = (1, 6, 11, 20, 29)


def seeded_vgg19_state_dict(seed=3):
    """A torchvision-STYLE ``vgg19`` state dict (keys ``features.N.weight|bias`` + a classifier entry that consumers must
    ignore) with seeded Kaiming weights: the ImageNet ones cannot be obtained offline, so parity tests inject
    the same seeded dict on both sides."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, (ci, co) in _VGG19_CONVS.items():
        w = torch.empty(co, ci, 3, 3)
        torch.nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu", generator=g)
        sd["features.%d.weight" % idx] = w
        sd["features.%d.bias" % idx] = torch.empty(co).uniform_(-0.1, 0.1, generator=g)
    sd["classifier.0.weight"] = torch.zeros(4, 4)
    return sd


class StockVGG19(torch.nn.Module):
    """``VGG19`` of ``models/networks/architecture.py:92-125`` on stock ops: torchvision's ``features`` indices 0..29
    (Conv2d 3x3 pad 1 / ReLU / MaxPool2d 2), outputs after relu1_1, relu2_1, relu3_1, relu4_1, relu5_1; weights frozen."""

    def __init__(self, state_dict):
        super().__init__()
        seq = []
        for i in range(30):
            if i in _VGG19_CONVS:
                conv = torch.nn.Conv2d(*_VGG19_CONVS[i], 3, padding=1)
                conv.weight.data.copy_(state_dict["features.%d.weight" % i])
                conv.bias.data.copy_(state_dict["features.%d.bias" % i])
                seq.append(conv)
            else:
                seq.append(torch.nn.MaxPool2d(2, 2) if i in _VGG19_POOLS else torch.nn.ReLU())
        self.features = torch.nn.Sequential(*seq)
        for q in self.parameters():
            q.requires_grad = False

    def forward(self, x):
        out = []
        for i, m in enumerate(self.features):
            x = m(x)
            if i in _VGG19_SLICE_ENDS:
                out.append(x)
        return out


@contextlib.contextmanager
def stock_sphere_ops():
    """Inside the block the product's SphereConv2D / SPADE norm + modulation run the restatements above."""
    from emlight_amd.GenProjector import spherenet
    saved = spherenet.sphere_conv, spherenet.spade_norm_modulate
    spherenet.sphere_conv, spherenet.spade_norm_modulate = sphere_conv, spade_norm_modulate
    try:
        yield
    finally:
        spherenet.sphere_conv, spherenet.spade_norm_modulate = saved
