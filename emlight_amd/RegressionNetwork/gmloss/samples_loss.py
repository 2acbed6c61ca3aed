"""``gmloss.SamplesLoss`` on the MI355X (reference ``RegressionNetwork/gmloss/samples_loss.py:12-84``).

``forward(x, y, geometry)``: ``geometry`` is the per-anchor depth; the reference rebuilds ``distance(batchsize,
geometry)`` on every call -- ``geometric_points`` (``gmloss/utils.py:63-74``) followed by an N^2 Python loop of
``torch.norm`` (``:76-92``, 16 384 tiny launches at N = 128) -- and then runs the same tensorised Sinkhorn as
``geomloss`` with cost ``(0.1 * |x_i - y_j|^2 + M_ij) / 2`` (``:94-108``, ``samples_loss.py:72-84``).

Here the ground cost is a *runtime argument* of the Sinkhorn kernel: the anchors are built in float64 on the host
(N numbers), ``eml_emd_anchor_cost_f32`` writes M on the device in one launch, and ``eml_sinkhorn_fwd_f32`` does the
rest -- the GMLight variant costs one extra 5 us launch per step.
"""
import numpy as np
import torch

from ... import _lib
from ..geomloss.samples_loss import SamplesLoss as _SphereSamplesLoss


def geometric_points(n=128, anchor_depth=None):
    """Depth-scaled Fibonacci anchors, float64 ``(n, 3)`` (``gmloss/utils.py:63-74``: radius = anchor_depth)."""
    golden_angle = np.pi * (3 - np.sqrt(5))
    theta = golden_angle * np.arange(n)
    z = np.linspace(1 - 1.0 / n, 1.0 / n - 1, n)
    radius = np.asarray(anchor_depth.detach().cpu().numpy() if isinstance(anchor_depth, torch.Tensor) else anchor_depth,
                        dtype=np.float64)
    points = np.zeros((n, 3))
    points[:, 0] = radius * np.cos(theta)
    points[:, 1] = radius * np.sin(theta)
    points[:, 2] = z
    return points


class SamplesLoss(_SphereSamplesLoss):
    """``SamplesLoss(loss="sinkhorn", p=2, blur=.05, reach=None, diameter=None, scaling=.5, batchsize=None)``,
    ``forward(x, y, geometry) -> (B,)`` with x, y of shape (B, 128, 1).  ``anchors`` lifts the reference's hard-coded
    N = 128 (``gmloss/utils.py:78``)."""

    def __init__(self, loss="sinkhorn", p=2, blur=.05, reach=None, diameter=None, scaling=.5, batchsize=None,
                 anchors=128):
        super().__init__(loss, p, blur, reach, diameter, scaling, batchsize, anchors=anchors)

    def forward(self, x, y, geometry):
        dev = x.device
        a = torch.from_numpy(geometric_points(self.N, geometry)).float().to(dev).contiguous()
        if not a.is_cuda:
            raise _lib.EmlightHipError("gmloss.SamplesLoss needs tensors on the MI355X; there is no CPU path")
        M = torch.empty(self.N, self.N, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().eml_emd_anchor_cost_f32(_lib.ptr(a), _lib.ptr(M), self.N, _lib.current_stream()),
                   "eml_emd_anchor_cost_f32")
        self.anchors = a
        self.M, self.Mt = M, M  # symmetric: the transpose aliases
        return super().forward(x, y)
