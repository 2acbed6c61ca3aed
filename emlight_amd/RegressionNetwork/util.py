"""Mirror of the hot-path parts of the reference's ``RegressionNetwork/util.py``.

``convert_to_panorama`` (reference ``util.py:222-245``) runs as one HIP kernel;
``sphere_points`` (``util.py:286-299``) and ``TonemapHDR`` (``util.py:36-66``) are the
small host helpers the kept entry points need.  EXR/vtk/cv2 I/O is out of scope.
"""
import numpy as np
import torch

from .. import _lib


def sphere_points(n=128):
    """Fibonacci-sphere anchors (n, 3), float64 -- reference ``util.py:286-299``."""
    idx = np.arange(n)
    ang = idx * (np.pi * (3.0 - np.sqrt(5.0)))
    z = np.linspace(1.0 - 1.0 / n, 1.0 / n - 1.0, n)
    rad = np.sqrt(1.0 - z * z)
    pts = np.empty((n, 3))
    pts[:, 0], pts[:, 1], pts[:, 2] = rad * np.cos(ang), rad * np.sin(ang), z
    return pts


class _Rasterise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dirs, sizes, colors, H, W):
        B, N = sizes.shape
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dirs.device)
        _lib.check(_lib.lib().eml_sg_rasterise_f32(_lib.ptr(dirs), _lib.ptr(sizes), _lib.ptr(colors),
                                                   _lib.ptr(out), B, N, H, W, _lib.current_stream()),
                   "eml_sg_rasterise_f32")
        ctx.save_for_backward(dirs, sizes)
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, gout):
        dirs, sizes = ctx.saved_tensors
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise NotImplementedError("convert_to_panorama: gradients wrt dirs/sizes are not on EMLight's "
                                      "path (anchors and lobe widths are constants)")
        H, W = ctx.hw
        B, N = sizes.shape
        return None, None, rasterise_bwd_colors_raw(dirs, sizes, gout, (H, W)), None, None


def rasterise_bwd_colors_raw(dirs, sizes, gout, pano_hw, exhaustive=False, legacy=False):
    """d loss / d colors (B, 3N) through ``eml_sg_rasterise_bwd_colors_ex_f32``: the forward's per-patch light lists, per-tile
    partial sums added in tile order.  ``exhaustive``: every light for every tile (what the culled launch equals bit for bit);
    ``legacy``: the round-1 kernel (one workgroup per 8 lights sweeping the whole panorama) -- the independent check."""
    H, W = pano_hw
    B, N = sizes.shape
    L = _lib.lib()
    gout = gout.contiguous()
    gcol = torch.empty(B, 3 * N, dtype=torch.float32, device=gout.device)
    if legacy:
        _lib.check(L.eml_sg_rasterise_bwd_colors_f32(_lib.ptr(dirs), _lib.ptr(sizes), _lib.ptr(gout), _lib.ptr(gcol), B, N, int(H),
                                                     int(W), _lib.current_stream()), "eml_sg_rasterise_bwd_colors_f32")
        return gcol
    work = torch.empty(max(1, L.eml_sg_rasterise_bwd_work_floats(B, N, int(H), int(W))), dtype=torch.float32, device=gout.device)
    _lib.check(L.eml_sg_rasterise_bwd_colors_ex_f32(_lib.ptr(dirs), _lib.ptr(sizes), _lib.ptr(gout), _lib.ptr(gcol), _lib.ptr(work),
                                                    B, N, int(H), int(W), 1 if exhaustive else 0, _lib.current_stream()),
               "eml_sg_rasterise_bwd_colors_ex_f32")
    return gcol


def rasterise_raw(dirs, sizes, colors, pano_hw=(128, 256), exhaustive=False, count=False):
    """``eml_sg_rasterise_ex_f32`` (no autograd): the panorama and, with ``count``, the number of exponentials the launch
    evaluated.  ``exhaustive``: every light for every pixel, in the reference's order (util.py:239-244) -- what the
    hierarchically culled default must equal bit for bit (tests); the bench reports both counts."""
    H, W = pano_hw
    d = _lib.require_gpu_tensor(dirs, "dirs")
    s = _lib.require_gpu_tensor(sizes, "sizes")
    c = _lib.require_gpu_tensor(colors, "colors")
    B, N = s.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=d.device)
    n = torch.zeros(1, dtype=torch.int64, device=d.device) if count else None
    _lib.check(_lib.lib().eml_sg_rasterise_ex_f32(_lib.ptr(d), _lib.ptr(s), _lib.ptr(c), _lib.ptr(out), B, N, int(H), int(W),
                                                  1 if exhaustive else 0, _lib.ptr(n), _lib.current_stream()),
               "eml_sg_rasterise_ex_f32")
    return (out, int(n.item())) if count else out


def convert_to_panorama(dirs, sizes, colors, pano_hw=(128, 256)):
    """SG lobes -> equirect panorama ``(B, 3, H, W)``; reference ``util.py:222-245``.

    dirs ``(B, 3N)``, sizes ``(B, N)``, colors ``(B, 3N)``.  ``pano_hw`` defaults to the
    reference's hard-coded 128 x 256 (``W`` must be ``2H``).  Differentiable wrt ``colors``.
    """
    H, W = pano_hw
    d = _lib.require_gpu_tensor(dirs, "dirs")
    s = _lib.require_gpu_tensor(sizes, "sizes")
    c = _lib.require_gpu_tensor(colors, "colors")
    B, N = s.shape
    if d.shape != (B, 3 * N) or c.shape != (B, 3 * N):
        raise ValueError("expected dirs (B,3N), sizes (B,N), colors (B,3N); got %s %s %s"
                         % (tuple(d.shape), tuple(s.shape), tuple(c.shape)))
    if B == 0:
        return c.new_zeros(0, 3, int(H), int(W))
    return _Rasterise.apply(d, s, c, int(H), int(W))


class TonemapHDR(object):
    """Global tonemap: alpha maps the ``percentile`` of I^(1/gamma) to ``max_mapping``
    (reference ``util.py:36-66``); numpy, host side (visualisation only)."""

    def __init__(self, gamma=2.4, percentile=50, max_mapping=0.5):
        self.gamma, self.percentile, self.max_mapping = gamma, percentile, max_mapping

    def __call__(self, numpy_img, clip=True, alpha=None, gamma=True):
        img = np.power(numpy_img, 1 / self.gamma) if gamma else numpy_img
        pos = img > 0
        ref = np.percentile(img[pos], self.percentile) if pos.any() else np.percentile(img, self.percentile)
        if alpha is None:
            alpha = self.max_mapping / (ref + 1e-10)
        out = np.multiply(alpha, img)
        if clip:
            out = np.clip(out, 0, 1)
        return out.astype("float32"), alpha
