"""Ground-truth parametrisation of HDR panoramas on the MI355X: ``extract_mesh`` of the reference's dataset
preparation (``RegressionNetwork/representation/distribution_representation.py:65-120``).

Same constructor and ``compute`` as the reference class, batched and on the device: ``compute(hdr)`` takes one
``(H, W, 3)`` panorama or a batch ``(B, H, W, 3)`` (float32, CUDA) and returns the dict the training pickles hold --
``distribution`` (N,), ``intensity`` (), ``rgb_ratio`` (3,), ``ambient`` (3,) (with a leading batch axis for a
batch) -- plus the lit-pixel map.  The reference loops over the N anchors in Python, masking the full panorama each
time; here the pixels are grouped by nearest anchor once (CSR) and one launch reduces every (image, anchor) cell in
float64.  Together with ``util.convert_to_panorama`` this closes the loop panorama -> parameters -> panorama.
"""
import torch

from .. import _lib
from .util import sphere_points


class extract_mesh:
    def __init__(self, h=128, w=256, ln=64, device="cuda"):
        self.h, self.w, self.ln = h, w, ln
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.EmlightHipError("extract_mesh runs on the MI355X; there is no CPU path")
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        self.anchors = torch.from_numpy(sphere_points(ln)).to(dev)                # float64, as the reference keeps them
        self.idx = torch.empty(h * w, dtype=torch.int32, device=dev)
        _lib.check(L.eml_gt_anchor_index_i32(p(self.anchors), ln, h, w, p(self.idx), st), "eml_gt_anchor_index_i32")
        # pixels grouped by anchor (CSR): makes the per-anchor sums a deterministic segmented reduction
        order = torch.argsort(self.idx.long(), stable=True)
        self.csr_pix = order.to(torch.int32).contiguous()
        self.csr_ptr = torch.zeros(ln + 1, dtype=torch.int32, device=dev)
        self.csr_ptr[1:] = torch.cumsum(torch.bincount(self.idx.long(), minlength=ln), 0).to(torch.int32)

    def compute(self, hdr):
        single = hdr.dim() == 3
        x = _lib.require_gpu_tensor(hdr.unsqueeze(0) if single else hdr, "hdr")
        B, H, W, C = x.shape
        if (H, W, C) != (self.h, self.w, 3):
            raise ValueError("expected hdr of shape (..., %d, %d, 3), got %s" % (self.h, self.w, tuple(hdr.shape)))
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        maxv = torch.empty(B, dtype=torch.float64, device=x.device)
        sums = torch.empty(B, self.ln + 1, 3, dtype=torch.float64, device=x.device)
        lit = torch.empty(B, H, W, dtype=torch.uint8, device=x.device)
        _lib.check(L.eml_gt_parametrise_f64(p(x), p(self.csr_ptr), p(self.csr_pix), B, H, W, self.ln, p(maxv), p(sums),
                                            p(lit), st), "eml_gt_parametrise_f64")
        anchors, ambient = sums[:, :self.ln], sums[:, self.ln]
        lum = torch.tensor([0.3, 0.59, 0.11], dtype=torch.float64, device=x.device)
        energy = anchors @ lum                                                    # :109
        rgb = anchors.sum(1)                                                      # :111
        intensity = torch.linalg.norm(rgb, dim=1)                                 # :112
        out = {"distribution": energy / energy.sum(1, keepdim=True), "intensity": intensity,
               "rgb_ratio": rgb / intensity[:, None], "ambient": ambient}
        lit = lit.bool().unsqueeze(-1)
        if single:
            return {k: v[0] for k, v in out.items()}, lit[0]
        return out, lit
