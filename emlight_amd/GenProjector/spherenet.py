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

from .._knobs import knob_choice, knob_flag, knob_int


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


def _require_gpu_f32(t, name):
    """No CPU path: the HIP engine takes f32 tensors on the GPU.  (Layout is handled by the callers: unlike
    ``_lib.require_gpu_tensor`` this does not force an NCHW-contiguous copy of a channels-last tensor.)"""
    from .. import _lib
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.EmlightHipError("%s must be a tensor on the MI355X (got %s); there is no CPU path"
                                   % (name, getattr(t, "device", type(t))))
    if t.dtype != torch.float32:
        raise _lib.EmlightHipError("%s must be float32 (got %s)" % (name, t.dtype))


class SphereGeometry:
    """Per (H, W, stride, device): the bilinear tap table of the sampling grid and its CSR transpose
    (``eml_sphere_tap_table_f32``; the transpose is what turns grid_sample's atomicAdd backward into a gather)."""

    def __init__(self, h, w, stride, device, kind="sphere"):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        self.h, self.w = h, w
        self.idx1 = self.wgt1 = None
        if kind == "planar":
            # an ordinary 3x3 convolution with zero padding 1 (the VGG19 feature stack of the perceptual loss,
            # architecture.py:92-125) as the degenerate case of the same gather: tap (a, b) of output pixel (r, c) is input
            # pixel (r*stride + a - 1, c*stride + b - 1) with weight 1 on the first corner, nothing on the other three
            self.ho, self.wo = (h + stride - 1) // stride, (w + stride - 1) // stride
            n = self.ho * self.wo * 9
            r = torch.arange(self.ho, device=device).view(-1, 1, 1, 1) * stride + torch.arange(3, device=device).view(1, 1, 3, 1) - 1
            c = torch.arange(self.wo, device=device).view(1, -1, 1, 1) * stride + torch.arange(3, device=device).view(1, 1, 1, 3) - 1
            ok = ((r >= 0) & (r < h) & (c >= 0) & (c < w)).reshape(n)
            src = (r * w + c).reshape(n)
            self.idx = torch.full((n, 4), -1, dtype=torch.int32, device=device)
            self.wgt = torch.zeros(n, 4, dtype=torch.float32, device=device)
            self.idx[:, 0] = torch.where(ok, src, torch.full_like(src, -1)).to(torch.int32)
            self.wgt[:, 0] = ok.float()
            # single-entry tables for the fused kernels (ke = 1): a quarter of the gather loads, no bilinear combine
            self.idx1, self.wgt1 = self.idx[:, 0].contiguous(), self.wgt[:, 0].contiguous()
        else:
            grid = sphere_sampling_grid(h, w, stride).to(device).contiguous()
            self.ho, self.wo = grid.shape[1] // 3, grid.shape[2] // 3
            n = self.ho * self.wo * 9
            self.idx = torch.empty(n, 4, dtype=torch.int32, device=device)
            self.wgt = torch.empty(n, 4, dtype=torch.float32, device=device)
            _lib.check(L.eml_sphere_tap_table_f32(p(grid), h, w, self.ho, self.wo, p(self.idx), p(self.wgt), st),
                       "eml_sphere_tap_table_f32")
        dst = self.idx.view(-1).long()
        keep = dst >= 0
        src = torch.arange(n, device=device).repeat_interleave(4)[keep]
        dst, wk = dst[keep], self.wgt.view(-1)[keep]
        order = torch.argsort(dst, stable=True)
        self.csr_src = src[order].to(torch.int32).contiguous()
        self.csr_w = wk[order].contiguous()
        self.csr_ptr = torch.zeros(h * w + 1, dtype=torch.int32, device=device)
        self.csr_ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=h * w), 0).to(torch.int32)
        self._transposed = None
        # EML_TAP_ROWSHARE (include/emlight_hip.h): on a stride-1 sphere grid a tap samples source column c + const(row, tap)
        # of two adjacent rows, so a pixel's east corners ARE its right neighbour's west corners -- verified on the table
        # itself (once per geometry), never assumed
        self.rowshare = int(kind == "sphere" and _table_rowshare(self.idx, self.wgt, self.ho * self.wo))

    def footprints(self, transposed=False):
        """For ``eml_sphere_conv_lowres_f32`` (csrc/gather_gemm3.h): per 128-pixel tile of a sample the contiguous range of SOURCE
        pixels its 9 taps touch -- ``(fp, fp_max, lidx)`` with ``fp`` an int32 (tiles, 2) tensor of (first pixel, count) or None
        when a tile spans whole samples (fewer than 128 destination pixels per sample), ``fp_max`` the largest count and ``lidx``
        the table's indices made footprint-local (``eml_sphere_conv_lowres_table_i32``); (None, 0, None) when the geometry does
        not tile.  ``transposed``: for the input gradient (destination = the input pixels, source = dY's)."""
        cache = self.__dict__.setdefault("_footprints", {})
        if transposed not in cache:
            if transposed:
                tt = self.transposed_table()
                idx, n_dst, n_src = (tt[0] if tt is not None else None), self.h * self.w, self.ho * self.wo
            else:
                idx, n_dst, n_src = self.idx, self.ho * self.wo, self.h * self.w
            fp, fp_max = None, 0
            if idx is None:
                pass
            elif n_dst % 128 == 0:
                t = idx.reshape(n_dst // 128, -1).long()
                ok = t >= 0
                lo = torch.where(ok, t, torch.full_like(t, n_src)).amin(1).clamp(max=n_src - 1)
                hi = torch.where(ok, t, torch.full_like(t, -1)).amax(1).clamp(min=0)
                n = (hi - lo + 1).clamp(min=1)
                fp, fp_max = torch.stack([lo, n], 1).to(torch.int32).contiguous(), int(n.max())
            elif n_dst < 128 and 128 % n_dst == 0:
                fp_max = (128 // n_dst) * n_src
            lidx = None
            if fp_max:
                from .. import _lib
                idx = idx.contiguous()
                lidx = torch.empty_like(idx)
                _lib.check(_lib.lib().eml_sphere_conv_lowres_table_i32(_lib.ptr(idx), _lib.ptr(fp), _lib.ptr(lidx), n_dst, idx.shape[-1],
                                                                       _lib.current_stream()), "eml_sphere_conv_lowres_table_i32")
            cache[transposed] = (fp, fp_max, lidx)
        return cache[transposed]

    def transposed_table(self):
        """The tap table seen from the INPUT pixels, for the fused input-gradient kernel: for input pixel q and tap t the
        (at most ke) output pixels whose tap t samples q with their bilinear weights, padded with (-1, 0).  Almost every
        (q, t) has exactly 4 entries at stride 1; the rows next to the poles have up to 8 (``rowmax`` tells the kernel
        which tiles need slots 4..7).  Returns None when a (q, t) has more than 8 entries (tiny geometries)."""
        if self._transposed is None:
            hw = self.h * self.w
            dst = self.idx.view(-1).long()
            keep = dst >= 0
            src = torch.arange(self.idx.shape[0], device=dst.device).repeat_interleave(4)[keep]
            key = dst[keep] * 9 + src % 9                       # (input pixel, tap)
            order = torch.argsort(key, stable=True)
            key, pix, wk = key[order], (src // 9)[order], self.wgt.view(-1)[keep][order]
            counts = torch.bincount(key, minlength=hw * 9)
            kmax = int(counts.max())
            if kmax > 8:
                self._transposed = (None,)
            else:
                ke = 4 if kmax <= 4 else 8
                first = torch.cumsum(counts, 0) - counts
                rank = torch.arange(key.numel(), device=key.device) - first[key]
                tidx = torch.full((hw * 9, ke), -1, dtype=torch.int32, device=key.device)
                twgt = torch.zeros(hw * 9, ke, dtype=torch.float32, device=key.device)
                tidx[key, rank] = pix.to(torch.int32)
                twgt[key, rank] = wk
                rowmax = counts.view(hw, 9).max(1).values.to(torch.uint8).contiguous()
                if kmax <= 1 and self.idx1 is not None:   # an ordinary convolution: one source per (pixel, tap)
                    tidx, twgt, ke = tidx[:, :1], twgt[:, :1], 1
                tidx, twgt = tidx.contiguous(), twgt.contiguous()
                self.t_rowshare = int(ke == 4 and self.idx1 is None and _table_rowshare(tidx, twgt, hw))
                self._transposed = (tidx, twgt, rowmax, ke)
        return self._transposed if self._transposed[0] is not None else None


def _table_rowshare(idx, wgt, n_dst):
    """Does a (n_dst * 9, 4) tap table have the property EML_TAP_ROWSHARE promises (include/emlight_hip.h)?  For every
    destination pixel p with p % 4 != 3 and every tap: entry 1 of p and entry 0 of p + 1 are the same source pixel unless one
    of them is -1 (grid_sample zero-pads the column that wraps around), likewise entries 3 / 2; -1 entries weigh 0."""
    if n_dst % 4 or idx.dim() != 2 or idx.shape[1] != 4 or idx.shape[0] != n_dst * 9:
        return False
    t = idx.view(n_dst // 4, 4, 9, 4)

    def same(a, b):
        return ((a == b) | (a < 0) | (b < 0)).all()
    ok = same(t[:, :3, :, 1], t[:, 1:, :, 0]) & same(t[:, :3, :, 3], t[:, 1:, :, 2]) & (wgt[idx < 0] == 0).all()
    return bool(ok)


_GEOMETRY = {}


def sphere_geometry(h, w, stride, device, kind="sphere"):
    key = (h, w, stride, str(device), kind)
    if key not in _GEOMETRY:
        _GEOMETRY[key] = SphereGeometry(h, w, stride, device, kind)
    return _GEOMETRY[key]


def _lowres_plan(geo, B, C, O, transposed=False):
    """Can ``eml_sphere_conv_lowres_f32`` run this product (C source channels -> O destination channels over ``geo``'s table, or
    its transposed one)?  -> (fp, fp_max, split) or None.  ``split``: channel chunks over that many workgroups per tile where the
    tiles alone would leave CUs idle (two resident workgroups per CU for the 256-thread variant, one for the 512-thread one)."""
    from .. import _lib
    L = _lib.lib()
    n_dst = geo.h * geo.w if transposed else geo.ho * geo.wo
    if B <= 0 or C % 32 or O % 128 or (n_dst % 128 and (n_dst > 128 or 128 % n_dst)):
        return None                      # (before the footprints: they cost a table pass and a host sync per geometry)
    fp, fp_max, lidx = geo.footprints(transposed)
    variant = L.eml_sphere_conv_lowres_variant(C, O, n_dst, fp_max) if fp_max else 0
    if not variant:
        return None
    tiles = ((B * n_dst + 127) // 128) * (O // (128 if variant == 1 else 256))
    resident, nch, split = (512 if variant == 1 else 256), C // 32, 1
    while tiles * split < resident and nch % (2 * split) == 0 and 8 * split * B * n_dst * O <= (256 << 20):
        split *= 2
    return fp, fp_max, split, lidx


def _lowres_conv(xr, idx, wgt, rowmax, ke, plan, w2, bias, res, slope, B, n_src, n_dst, C, O):
    """One call of ``eml_sphere_conv_lowres_f32``: (B * n_dst, O) = act(gather(xr) W2^T + bias + res).  (``idx``: the raw table the
    plan's footprint-local one was made from -- kept in the signature for the callers' symmetry, the kernel takes the plan's.)"""
    from .. import _lib
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    fp, fp_max, split, idx = plan
    y = torch.empty(B * n_dst, O, dtype=torch.float32, device=xr.device)
    part = (torch.empty(L.eml_sphere_conv_lowres_partial_floats(B * n_dst, O, split), dtype=torch.float32, device=xr.device)
            if split > 1 else None)
    _lib.check(L.eml_sphere_conv_lowres_f32(p(xr), p(idx), p(wgt), p(rowmax), ke, p(fp), fp_max, p(w2), p(bias), p(y), p(part), split,
                                            B, n_src, n_dst, C, O, p(res), float(slope), st), "eml_sphere_conv_lowres_f32")
    return y


class _SphereConvFn(torch.autograd.Function):
    """``conv2d(grid_sample(x, grid), weight, bias, stride=3)`` (``sphere_cnn.py:121-124``).

    Large layers (by the size of the would-be im2col operand ``A9``, thresholds in ``forward``; channel counts that
    tile): the FUSED kernels of ``csrc/sphere_conv_fused.hip`` -- taps gathered straight into the LDS operand of an
    f32-MFMA implicit GEMM: forward (with the consumer's residual sum / activation in the epilogue), weight gradient, and
    the input gradient by the forward kernel on the transposed tap table, so the 9x blown-up operand never exists in HBM.
    The 3-channel input layers (3 -> 64 / 128): the one-pass kernels of ``csrc/sphere_conv_small.hip``.  Everything else
    (wide low-resolution heads, odd channel counts): im2col_sphere (HIP) -> library GEMM, and for the input gradient
    ``dY W2`` (library GEMM) -> col2im_sphere (HIP, deterministic gather over the CSR transpose of the tap table).
    ``weight`` may be any (O, C, 3, 3) tensor; one whose memory is already (O, 3, 3, C) -- the fused spectral norm's
    result -- is used without a re-layout copy, and the weight gradient is returned in that memory order.
    Activations are pixel-major: inputs in ``torch.channels_last`` are used in place, the output is returned as a
    channels-last (B, O, H', W') tensor, so a chain of SphereConvs never transposes."""

    @staticmethod
    def _im2col(xr, geo, B, C):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        po = geo.ho * geo.wo
        a9 = torch.empty(B * po, 9 * C, dtype=torch.float32, device=xr.device)
        _lib.check(L.eml_sphere_im2col_f32(p(xr), p(geo.idx), p(geo.wgt), p(a9), B, geo.h * geo.w, po, C, st),
                   "eml_sphere_im2col_f32")
        return a9

    @staticmethod
    def forward(ctx, x, weight, bias, stride, kind="sphere", residual=None, slope=1.0):
        """``residual`` (B, O, H', W') and ``slope``: the consumer's epilogue folded in, ``leaky_relu(conv + bias + residual,
        slope)`` (slope 1 = none, 0 = ReLU) -- the ``x_s + dx`` of a SPADEResnetBlock (architecture.py:60), VGG's ReLUs."""
        from .. import _lib
        _require_gpu_f32(x, "SphereConv2D input")
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        B, C, H, W = x.shape
        geo = sphere_geometry(H, W, stride, x.device, kind)
        po = geo.ho * geo.wo
        xr = x.permute(0, 2, 3, 1).contiguous()                   # (B,H,W,C); a view when x is channels-last
        O = weight.shape[0]
        w2 = _frozen_layout(weight, "w2", kind == "planar")       # columns ordered (tap, c) like A9
        # which layers take the fused kernels: measured per shape (tools/sphere_layers.py, profiles/r02_sphere_layers.jsonl).
        # They run at 85-115 TF/s; im2col + the library GEMM is faster only where the GEMM is wide (O >= 512) and the
        # operand small -- there the library's 115-125 TF/s wins and A9 costs little memory.
        a9_bytes = B * po * 9 * C * 4
        lim = SphereConv2D.fused_min_bytes
        ctx.fused_fwd = (B > 0 and C % 32 == 0 and O % 64 == 0 and
                         (a9_bytes >= 32 * lim or (O <= 256 and a9_bytes >= lim)))
        ctx.fused_dgrad, ctx.fused_wgrad = _backward_dispatch(B, po, C, O, stride)
        a9 = None
        slope = float(slope)
        res = None
        if residual is not None:
            if residual.shape != (B, O, geo.ho, geo.wo):
                raise ValueError("SphereConv2D residual %s does not match the output (%d, %d, %d, %d)"
                                 % (tuple(residual.shape), B, O, geo.ho, geo.wo))
            res = residual.permute(0, 2, 3, 1).contiguous().view(B * po, O)    # a view when it is channels-last
        # 3-channel input layers (SPADE's mlp_shared 3 -> 128, VGG's 3 -> 64; not the discriminator's 6 -> 64): bound by the
        # write of their output -- one pass each way with the activation (and its backward, and the bias gradient) folded in
        ctx.small = bool(B > 0 and res is None and L.eml_sphere_conv_small_supported(C, O))
        # few-channel OUTPUT layers (conv_img 64 -> 3, the discriminators' final convolutions): one-pass kernels, no 9x operand
        ctx.narrow = bool(SphereConv2D.narrow_kernels and B > 0 and res is None and slope == 1.0 and geo.idx1 is None
                          and not ctx.small and L.eml_sphere_conv_narrow_supported(C, O))
        # round 6: the low-resolution, wide layers (everything that used to fall through to im2col + a library GEMM below) on
        # the footprint gather-GEMM of csrc/gather_gemm3.h; "force": wherever it is supported (tests, A/B), "off": round 5
        mode = SphereConv2D.lowres
        plan = None
        if mode != "off" and kind == "sphere" and stride == 1 and not ctx.small and not ctx.narrow and (mode == "force" or not ctx.fused_fwd):
            plan = _lowres_plan(geo, B, C, O)
        ctx.lowres_dgrad = None
        if mode != "off" and kind == "sphere" and stride == 1 and not ctx.small and not ctx.narrow and (mode == "force" or not ctx.fused_dgrad):
            ctx.lowres_dgrad = _lowres_plan(geo, B, O, C, transposed=True) if ctx.needs_input_grad[0] else None
        if ctx.narrow:
            ctx.fused_fwd = ctx.fused_wgrad = ctx.fused_dgrad = False
            y = torch.empty(B * po, O, dtype=torch.float32, device=x.device)
            if SphereConv2D.narrow_project:   # round 6: project onto the 9 x O columns per SOURCE pixel, then gather 16 bytes per corner
                scratch = torch.empty(L.eml_sphere_conv_narrow_scratch_floats(B, H * W), dtype=torch.float32, device=x.device)
                _lib.check(L.eml_sphere_conv_narrow_fwd2_f32(p(xr), p(geo.idx), p(geo.wgt), p(w2.contiguous()),
                                                             p(bias.contiguous()) if bias is not None else None, p(y), p(scratch),
                                                             B, H * W, po, C, O, st), "eml_sphere_conv_narrow_fwd2_f32")
                del scratch
            else:
                _lib.check(L.eml_sphere_conv_narrow_fwd_f32(p(xr), p(geo.idx), p(geo.wgt), p(w2.contiguous()),
                                                            p(bias.contiguous()) if bias is not None else None, p(y), B, H * W, po,
                                                            C, O, st), "eml_sphere_conv_narrow_fwd_f32")
        elif ctx.small:
            ctx.fused_fwd = ctx.fused_wgrad = False
            y = torch.empty(B * po, O, dtype=torch.float32, device=x.device)
            _lib.check(L.eml_sphere_conv_small_fwd_f32(p(xr), p(geo.idx), p(geo.wgt), p(w2.contiguous()),
                                                       p(bias.contiguous()) if bias is not None else None, p(y), B, H * W,
                                                       po, C, O, slope, st), "eml_sphere_conv_small_fwd_f32")
        elif plan is not None:
            y = _lowres_conv(xr, geo.idx, geo.wgt, None, 4, plan, w2.contiguous(), bias.contiguous() if bias is not None else None,
                             res, slope, B, H * W, po, C, O)
        elif ctx.fused_fwd:
            y = torch.empty(B * po, O, dtype=torch.float32, device=x.device)
            tab = (geo.idx1, geo.wgt1, 1) if geo.idx1 is not None else (geo.idx, geo.wgt, 4)
            _lib.check(L.eml_sphere_conv_fwd_fused_ex_f32(p(xr), p(tab[0]), p(tab[1]), p(w2.contiguous()),
                                                          p(bias.contiguous()) if bias is not None else None, p(y), B,
                                                          H * W, po, C, O, tab[2], p(res) if res is not None else None,
                                                          slope, geo.rowshare if tab[2] == 4 else 0, st),
                       "eml_sphere_conv_fwd_fused_ex_f32")
        else:
            a9 = _SphereConvFn._im2col(xr, geo, B, C) if B else xr.new_empty(0, 9 * C)
            y = torch.addmm(bias, a9, w2.t()) if bias is not None else a9 @ w2.t()
            if res is not None:
                y += res
            if slope != 1.0:
                y = torch.relu_(y) if slope == 0.0 else nn.functional.leaky_relu_(y, slope)
        # A9 exists (the forward ran as im2col + library GEMM: wide heads, O >= 512): the weight gradient is then ONE long-K
        # library GEMM on it -- 145-152 TF/s on the recorded selection (_gemm_selection.py) against the fused kernel's 110,
        # with no gather pass of its own.  (Round 5; before the GEMMs were tuned the fused kernel won above 256 MB of operand.)
        if a9 is not None and SphereConv2D.lib_wgrad_on_kept_operand and SphereConv2D.keep_operand and weight.requires_grad:
            ctx.fused_wgrad = False
        # the library weight gradient needs A9 again: keep it (9x the input) or rebuild it from x; the fused one never does
        ctx.keep = (a9 is not None and not ctx.fused_wgrad and SphereConv2D.keep_operand and weight.requires_grad)
        ctx.slope = slope
        # an activation's derivative is taken from the OUTPUT's sign (valid for slope >= 0; the entry point rejects others)
        ctx.save_for_backward(a9 if ctx.keep else xr, weight, *((y,) if slope != 1.0 else ()))
        ctx.geo, ctx.has_bias, ctx.shape = geo, bias is not None, (B, C, H, W, O)
        return y.view(B, geo.ho, geo.wo, O).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        return _sphere_conv_backward(ctx, gy, ctx.saved_tensors, ctx.needs_input_grad)


def _backward_dispatch(B, po, C, O, stride):
    """(fused input gradient?, fused weight gradient?) of a SphereConv layer, measured per shape (tools/sphere_layers.py)."""
    lim = SphereConv2D.fused_min_bytes
    a9_bytes = B * po * 9 * C * 4
    # input gradient: the forward kernel on the transposed tap table (K = 9*O, N = C).  Measured: it beats
    # dY W2 (library) + col2im where the pixel count is large and O <= 256; wide heads (K = 9*O >= 4608) stay unfused
    fused_dgrad = (B > 0 and O % 32 == 0 and C % 64 == 0 and stride == 1 and
                   (lim == 0 or (O <= 256 and B * po >= 131072)))
    # weight gradient: K = pixels; below ~32k pixels the split-K tiles are short and the library's long-K GEMM wins
    fused_wgrad = (B > 0 and C % 64 == 0 and O >= 64 and O % 16 == 0 and a9_bytes >= 4 * lim and
                   (B * po >= 32768 or lim == 0))
    return fused_dgrad, fused_wgrad


def _sphere_conv_backward(ctx, gy, saved, needs):
    """Backward of ``_SphereConvFn`` given its recorded state ``ctx`` (geo, shape, slope, dispatch flags), the saved tensors and
    the needs-gradient flags in ``forward``'s argument order; shared with the fused SPADE forward, whose backward feeds it the
    (dgamma | dbeta) tensor."""
    from .. import _lib
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    xr, weight = saved[:2]
    geo = ctx.geo
    B, C, H, W, O = ctx.shape
    po = geo.ho * geo.wo
    gyr = gy.permute(0, 2, 3, 1).reshape(B * po, O).contiguous()
    y = saved[2] if ctx.slope != 1.0 else None
    gx = gw = gb = gres = None
    # the 3-channel input layers: dW2, the bias gradient and the activation's backward in one pass over (dY, Y); where the
    # input itself carries a gradient (the guide map of the joint step, VGG's conv1_1 on the generated panorama) the masked
    # dY times W2 in one more pass (eml_sphere_conv_small_da9_f32) -- no activation-backward pass, no N = 27 library GEMM
    small_w = ctx.small and needs[1]
    small_x = ctx.small and needs[0] and B > 0 and SphereConv2D.small_input_grad
    if ctx.small and needs[0] and not small_x:
        small_w = False   # EML_SMALL_DA9=0 (A/B): round 4's dispatch -- the general path forms the masked dY once for both
    if small_w:
        part = torch.empty(L.eml_sphere_conv_small_wgrad_partial_floats(B, po, C, O), dtype=torch.float32, device=gy.device)
        gw2 = torch.empty(O, 9 * C, dtype=torch.float32, device=gy.device)
        gb_ = torch.empty(O, dtype=torch.float32, device=gy.device) if ctx.has_bias else None
        _lib.check(L.eml_sphere_conv_small_wgrad_f32(p(xr), p(geo.idx), p(geo.wgt), p(gyr), p(y) if y is not None else None,
                                                     ctx.slope, p(part), p(gw2), p(gb_) if gb_ is not None else None, B,
                                                     H * W, po, C, O, st), "eml_sphere_conv_small_wgrad_f32")
        gw = gw2.view(O, 3, 3, C).permute(0, 3, 1, 2)
        gb = gb_ if (ctx.has_bias and needs[2]) else None
        del part
    if small_x:
        w2 = weight.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
        da9 = torch.empty(B * po, 9 * C, dtype=torch.float32, device=gy.device)
        _lib.check(L.eml_sphere_conv_small_da9_f32(p(gyr), p(y) if y is not None else None, ctx.slope, p(w2), p(da9), B * po, C, O,
                                                   st), "eml_sphere_conv_small_da9_f32")
        gxr = torch.empty(B, H, W, C, dtype=torch.float32, device=gy.device)
        _lib.check(L.eml_sphere_col2im_f32(p(da9), p(geo.csr_ptr), p(geo.csr_src), p(geo.csr_w), p(gxr), B, H * W, po, C, st),
                   "eml_sphere_col2im_f32")
        gx = gxr.permute(0, 3, 1, 2)
    need_masked = ((needs[0] and not small_x) or (len(needs) > 5 and needs[5])
                   or (not small_w and (needs[1] or (ctx.has_bias and needs[2]))))
    if y is not None and need_masked:
        gyr = (torch.ops.aten.threshold_backward(gyr, y, 0.0) if ctx.slope == 0.0
               else torch.ops.aten.leaky_relu_backward(gyr, y, ctx.slope, True))
    if len(needs) > 5 and needs[5]:
        gres = gyr.view(B, geo.ho, geo.wo, O).permute(0, 3, 1, 2)
    if ctx.has_bias and needs[2] and not small_w:
        # the SPADE modulation's backward leaves the column sums of the dgb it produced on the tensor (f64-accumulated)
        # -- valid only for the very tensor it was computed from: same storage, not written since (autograd accumulates
        # IN PLACE into a gradient that has a second consumer; a hook may hand over a different tensor): ADVICE round 3
        pre = getattr(gy, "_eml_colsum", None)
        if pre is not None:
            sums, ptr, ver = pre
            ok = y is None and sums.shape == (O,) and ptr == gy.data_ptr() and ver == gy._version
            pre = sums if ok else None
        if pre is not None:
            gb = pre
        elif O <= 8 and gyr.is_cuda and B > 0:   # tall and skinny: ATen's sum(0) runs it on a few workgroups
            part = torch.empty(L.eml_colsum_partial_doubles(O), dtype=torch.float64, device=gy.device)
            gb = torch.empty(O, dtype=torch.float32, device=gy.device)
            _lib.check(L.eml_colsum_f32(p(gyr), B * po, O, p(part), p(gb), st), "eml_colsum_f32")
        else:
            gb = gyr.sum(0)
        if gres is not None and gb is not None:
            # the residual branch's gradient IS this (masked) dY: a learned shortcut's bias gradient (conv_s, architecture.py:60)
            # is the same column sum -- handed over on the tensor, under the same storage / version check as above
            gres._eml_colsum = (gb, gres.data_ptr(), gres._version)
    narrow = getattr(ctx, "narrow", False)
    narrow_v = None   # the few-output-channel layers' V (B*HW, 36), left by the weight gradient for the input gradient
    if needs[1] and not small_w:
        ttn = geo.transposed_table() if (narrow and SphereConv2D.narrow_project and B) else None
        if narrow and ttn is not None:
            # round 6: what every output pixel sends back to a source pixel through each tap (the transposed table), (B*HW, 36),
            # then a 36 x C product over the pixels -- nothing is gathered at full channel width
            tidx, twgt, rowmax, ke = ttn
            scratch = torch.empty(L.eml_sphere_conv_narrow_scratch_floats(B, H * W), dtype=torch.float32, device=gy.device)
            part = torch.empty(L.eml_sphere_conv_narrow_wgrad2_partial_floats(B, H * W, C), dtype=torch.float32, device=gy.device)
            gw2 = torch.empty(O, 9 * C, dtype=torch.float32, device=gy.device)
            _lib.check(L.eml_sphere_conv_narrow_wgrad2_f32(p(xr), p(tidx), p(twgt), ke, p(rowmax) if ke == 8 else None, p(gyr),
                                                           p(scratch), p(part), p(gw2), B, H * W, po, C, O, st),
                       "eml_sphere_conv_narrow_wgrad2_f32")
            gw = gw2.view(O, 3, 3, C).permute(0, 3, 1, 2)
            del part
            narrow_v = scratch     # V stays for the input gradient below
        elif narrow:
            part = torch.empty(L.eml_sphere_conv_narrow_wgrad_partial_floats(B, po, C, O), dtype=torch.float32, device=gy.device)
            gw2 = torch.empty(O, 9 * C, dtype=torch.float32, device=gy.device)
            _lib.check(L.eml_sphere_conv_narrow_wgrad_f32(p(xr), p(geo.idx), p(geo.wgt), p(gyr), p(part), p(gw2), B, H * W, po, C,
                                                          O, st), "eml_sphere_conv_narrow_wgrad_f32")
            gw = gw2.view(O, 3, 3, C).permute(0, 3, 1, 2)
            del part
        elif ctx.fused_wgrad and B:
            bn = 128 if C % 128 == 0 else 64
            bmo = 128 if (O % 128 == 0 or O > 192) else 64
            tiles = 9 * (C // bn) * ((O + bmo - 1) // bmo)
            nchunks = (B * po + 31) // 32
            split = max(1, min(nchunks, SphereConv2D.wgrad_workgroups // tiles, (512 << 20) // (O * 9 * C * 4)))
            part = torch.empty(L.eml_sphere_conv_wgrad_partial_floats(C, O, split), dtype=torch.float32, device=gy.device)
            gw2 = torch.empty(O, 9 * C, dtype=torch.float32, device=gy.device)
            _lib.check(L.eml_sphere_conv_wgrad_fused_f32(p(xr), p(geo.idx), p(geo.wgt), p(gyr), p(part), p(gw2), B,
                                                         H * W, po, C, O, split, st), "eml_sphere_conv_wgrad_fused_f32")
            gw = gw2.view(O, 3, 3, C).permute(0, 3, 1, 2)
            del part
        else:
            a9 = xr if ctx.keep else _SphereConvFn._im2col(xr, geo, B, C)
            m_rows, split = B * po, 1
            if (9 * C <= 64 or O <= 16) and m_rows >= 65536:
                # skinny product (the 3- / 6-channel input layers: 27 or 54 columns against ~1 M rows; the 3-channel
                # output layers: 1.9 TF/s as one GEMM): as ONE GEMM
                # the library runs it on a handful of workgroups (1.6 ms for 7 GFLOP); as a batched split-K it is
                # the HBM-bound read of dY it should be, followed by a fixed-order sum of the partials
                split = 512
                while m_rows % split:
                    split //= 2
            if split > 1:
                gw2 = torch.bmm(a9.view(split, m_rows // split, 9 * C).transpose(1, 2),
                                gyr.view(split, m_rows // split, O)).sum(0)
                gw = gw2.t().contiguous().view(O, 3, 3, C).permute(0, 3, 1, 2)
            else:
                # (9C, O) = A9^T gy, then one transposing copy into the (O, tap, c) layout of every other weight gradient.
                # (gy^T A9 gives that layout directly: measured in round 5 on the recorded GEMM selection -- the copies it saves,
                # 0.5 ms per joint step, are what that orientation's GEMMs lose: profiles/r05_ab_dispatch_late.txt)
                gw2 = a9.t() @ gyr
                gw = gw2.t().contiguous().view(O, 3, 3, C).permute(0, 3, 1, 2)
            del a9
    if needs[0] and not small_x:
        lowres = getattr(ctx, "lowres_dgrad", None)
        gxr = torch.empty(B, H, W, C, dtype=torch.float32, device=gy.device) if (lowres is None or narrow or not B) else None
        tt = geo.transposed_table() if ((ctx.fused_dgrad or narrow or lowres is not None) and B) else None
        if tt is not None and lowres is not None and not narrow:
            # the footprint gather-GEMM over the transposed table (round 6): dY's rows of the few source rows in LDS
            tidx, twgt, rowmax, ke = tt
            w2t = weight.permute(1, 2, 3, 0).reshape(C, 9 * O).contiguous()   # columns ordered (tap, o)
            gxr = _lowres_conv(gyr, tidx, twgt, rowmax if ke == 8 else None, ke, lowres, w2t, None, None, 1.0, B, po, H * W, O,
                               C).view(B, H, W, C)
        elif tt is not None and narrow:
            tidx, twgt, rowmax, ke = tt
            w2 = weight.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
            # round 6: dX = V W2 on the matrix unit, V shared with the weight gradient -- for conv_img's width (372 -> 86 us); the
            # discriminators' 512-channel heads keep the one-pass kernel (a workgroup would stage 74 KB of weights for a few
            # thousand pixels: 35 -> 31 us at 15 x 31, 12 -> 20 us at 7 x 15; tools/bench_narrow.py)
            if SphereConv2D.narrow_project and C <= 128:
                has_v = narrow_v is not None
                scratch = narrow_v if has_v else torch.empty(L.eml_sphere_conv_narrow_scratch_floats(B, H * W), dtype=torch.float32,
                                                             device=gy.device)
                _lib.check(L.eml_sphere_conv_narrow_dgrad2_f32(p(gyr), p(tidx), p(twgt), ke, p(rowmax) if ke == 8 else None, p(w2),
                                                               p(gxr), p(scratch), int(has_v), B, H * W, po, C, O, st),
                           "eml_sphere_conv_narrow_dgrad2_f32")
                del scratch
            else:
                _lib.check(L.eml_sphere_conv_narrow_dgrad_f32(p(gyr), p(tidx), p(twgt), ke, p(w2), p(gxr), B, H * W, po, C, O, st),
                           "eml_sphere_conv_narrow_dgrad_f32")
        elif tt is not None:
            # gather-GEMM over the transposed tap table: neither dA9 (B*Po, 9C) nor its col2im pass exist
            tidx, twgt, rowmax, ke = tt
            w2t = _frozen_layout(weight, "w2t", geo.idx1 is not None)          # columns ordered (tap, o)
            _lib.check(L.eml_sphere_conv_dgrad_fused_f32(p(gyr), p(tidx), p(twgt), p(rowmax), ke, p(w2t), p(gxr), B,
                                                         H * W, po, C, O, getattr(geo, "t_rowshare", 0), st),
                       "eml_sphere_conv_dgrad_fused_f32")
        else:
            w2 = weight.permute(0, 2, 3, 1).reshape(O, 9 * C)
            da9 = gyr @ w2                                           # (B*Po, 9C)
            if B:
                _lib.check(L.eml_sphere_col2im_f32(p(da9), p(geo.csr_ptr), p(geo.csr_src), p(geo.csr_w), p(gxr), B,
                                                   H * W, po, C, st), "eml_sphere_col2im_f32")
        gx = gxr.permute(0, 3, 1, 2)
    if gw is not None and weight.is_leaf:
        # data-parallel runs, a weight that is a parameter itself (SPADE's heads, conv_img, the first discriminator layer):
        # the re-layout to the parameter's own order -- which autograd would make anyway when it adopts a permuted view --
        # lands in the parameter's place in its gradient bucket (_dist.grad_slot: nothing to pack)
        from .._dist import grad_slot
        slot = grad_slot(ptr=weight.data_ptr())
        if slot is not None and slot.shape == gw.shape and slot.device == gw.device:
            slot.copy_(gw)
            gw = slot
    return gx, gw, gb, None, None, gres, None


def sphere_conv(x, weight, bias, stride=1, residual=None, act_slope=1.0):
    """``conv2d(grid_sample(x, grid(H, W, stride)), weight, bias, stride=3)`` on the MI355X (the one execution path;
    tests swap this attribute for the oracle's stock-op restatement when they need a CPU run).  ``residual`` / ``act_slope``:
    ``leaky_relu(conv + residual, act_slope)`` in the producing kernel's epilogue (defaults: the plain convolution)."""
    if residual is None and act_slope == 1.0:
        return _SphereConvFn.apply(x, weight, bias, stride)
    return _SphereConvFn.apply(x, weight, bias, stride, "sphere", residual, act_slope)


_FROZEN = {}   # (id, kind) -> (weakref to the Parameter, its version, re-laid-out copy): frozen Parameters only


def _frozen_layout(weight, kind, remember=False):
    """``weight`` (O, C, 3, 3) as the (O, 9C) operand of the forward ("w2": columns (tap, c)) or the (C, 9O) operand of the
    fused input gradient ("w2t": columns (tap, o)).  A trainable weight changes every step and is re-laid-out per call (a
    view here, made contiguous where it is used); a FROZEN one -- the VGG19 stack of the perceptual loss: 13 convolutions, two
    forward passes and one input-gradient pass per iteration, up to 9.4 MB each -- is copied once per (object, version, storage);
    a frozen weight rewritten through ``.data`` in place is not seen (no version bump): write it under ``torch.no_grad()``."""
    O, C = weight.shape[0], weight.shape[1]
    view = (weight.permute(0, 2, 3, 1).reshape(O, 9 * C) if kind == "w2" else weight.permute(1, 2, 3, 0).reshape(C, 9 * O))
    # only a Parameter object that takes no gradient is remembered, and only while it is THAT object at THAT version (a
    # computed weight -- spectral norm under no_grad -- is a new tensor every call whose address the allocator reuses)
    # ... and only for the planar convolutions (``remember``: the VGG19 stack, which no optimizer holds).  A discriminator frozen
    # for the generator step is a Parameter without requires_grad too, and the fused Adam that updates it a moment later does
    # not bump its version counter: remembered by version, its head kept last step's weights (the joint reproducibility test
    # caught it: GAN term off by 0.017).
    if not remember or weight.requires_grad or not isinstance(weight, torch.nn.Parameter):
        return view if kind == "w2" else view.contiguous()
    key = (id(weight), kind)
    hit = _FROZEN.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != (weight._version, weight.data_ptr()) or hit[2].device != weight.device:
        if len(_FROZEN) > 256:
            _FROZEN.clear()
        import weakref
        hit = _FROZEN[key] = (weakref.ref(weight), (weight._version, weight.data_ptr()), view.contiguous())
    return hit[2]


def planar_conv3x3(x, weight, bias, stride=1, act_slope=1.0):
    """``F.conv2d(x, weight, bias, stride, padding=1)`` for a 3x3 kernel through the same gather + f32-MFMA kernels as
    SphereConv2D (the tap table of an ordinary convolution; used by the VGG19 feature stack of the perceptual loss).
    ``act_slope`` = 0: the ReLU that follows every VGG convolution, applied in the epilogue."""
    return _SphereConvFn.apply(x, weight, bias, stride, "planar", None, act_slope)


def _rows_view(t):
    """(B,C,H,W) tensor in channels-last memory -> (its (B*H*W, C) row-major alias, row stride)."""
    b, c, h, w = t.shape
    if t.stride(1) != 1 or t.stride(3) % 4 or t.stride(2) != w * t.stride(3) or (b > 1 and t.stride(0) != h * t.stride(2)):
        t = t.contiguous(memory_format=torch.channels_last)
    return t, t.stride(3)


class _SpadeModulateFn(torch.autograd.Function):
    """``leaky_relu(normalized * (1 + gamma) + beta, slope)`` (``normalization.py:113-115`` + ``architecture.py:56-57``)
    with gamma | beta taken from the two channel halves of one tensor ``gb`` (B, 2C, H, W): one HIP pass forward,
    one backward (``eml_spade_modulate_*``)."""

    @staticmethod
    def forward(ctx, xn, gb, slope):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        _require_gpu_f32(xn, "SPADE normalized input")
        B, C, H, W = xn.shape
        xn, ldx = _rows_view(xn)
        gb, ldg = _rows_view(gb)
        y = torch.empty_like(xn, memory_format=torch.channels_last)
        _lib.check(L.eml_spade_modulate_fwd_f32(p(xn), ldx, p(gb), ldg, p(y), C, B * H * W, C, float(slope), st),
                   "eml_spade_modulate_fwd_f32")
        ctx.save_for_backward(xn, gb)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        xn, gb = ctx.saved_tensors
        B, C, H, W = xn.shape
        gy, ldy = _rows_view(gy)
        dxn = torch.empty_like(xn, memory_format=torch.channels_last)
        dgb = torch.empty((B, 2 * C, H, W), dtype=torch.float32, device=xn.device, memory_format=torch.channels_last)
        _lib.check(L.eml_spade_modulate_bwd_f32(p(gy), ldy, p(xn), xn.stride(3), p(gb), gb.stride(3), p(dxn), C, p(dgb),
                                                2 * C, B * H * W, C, ctx.slope, st), "eml_spade_modulate_bwd_f32")
        return dxn, dgb, None


def _stats_grid(rows, C):
    cvp = 1
    while cvp < C // 4 and cvp < 256:
        cvp <<= 1
    rpp = 256 // cvp
    return max(1, min(512, (rows + rpp - 1) // rpp))


def _bn_sync():
    """SPADE's norm is SYNCHRONISED batch norm in the reference (sync_batchnorm/batchnorm.py:105-126, across the
    DataParallel replicas); here replicas are processes: the (2C+1)-float sums are all-reduced over RCCL."""
    from .._dist import dp_active
    return dp_active()


def _reduce_sums(partials, rows, C, repeat=1):
    """[grid][C][2] f64 partials -> sums (2C+1,) f64 = per-channel pairs then the row count; all-reduced across ranks.
    ``repeat``: every row counts that many times (statistics of a nearest-upsampled map taken from the map itself)."""
    from .. import _lib
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=partials.device)
    sums[2 * C:].fill_(float(rows))   # (a fill kernel: `t[i] = python_float` is a host->device copy, which a stream capture refuses)
    _lib.check(L.eml_bn_fold_f64(p(partials), partials.shape[0], 2 * C, p(sums), st), "eml_bn_fold_f64")
    if repeat != 1:
        sums *= float(repeat)
    if _bn_sync():
        import torch.distributed as dist
        dist.all_reduce(sums)
    return sums


def spade_batch_stats(x, bn, also=(), repeat=1):
    """(mean, istd) of SPADE's parameter-free BatchNorm for input ``x`` (normalization.py:101-104): batch statistics in
    training (one read of x, f64 accumulation; running statistics of ``bn`` updated like nn.BatchNorm2d), running
    statistics in eval.  Not differentiable: the statistics' gradient is part of ``spade_norm_modulate``'s backward.
    ``also``: further parameter-free BatchNorms over the SAME x (SPADEResnetBlock's norm_s next to norm_0): their running
    buffers are updated from the same sums, each with its own momentum and its own previous values."""
    from .. import _lib
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    _require_gpu_f32(x, "SPADE input")
    B, C, H, W = x.shape
    if not bn.training:
        return bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
    with torch.no_grad():
        xr, ld = _rows_view(x)
        rows = B * H * W
        grid = _stats_grid(rows, C)
        partials = torch.empty(grid, C, 2, dtype=torch.float64, device=x.device)
        _lib.check(L.eml_bn_stats_f32(p(xr), ld, rows, C, p(partials), grid, st), "eml_bn_stats_f32")
        sums = _reduce_sums(partials, rows, C, repeat)   # repeat = 4: x stands for its nearest x2 upsample
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        istd = torch.empty(C, dtype=torch.float32, device=x.device)
        mom = bn.momentum if bn.momentum is not None else 0.1
        _lib.check(L.eml_bn_finalize_f32(p(sums), C, float(bn.eps), float(mom), p(mean), p(istd), p(bn.running_mean),
                                         p(bn.running_var), st), "eml_bn_finalize_f32")
        bn.num_batches_tracked += 1
        for other in also:
            if other.training:
                m2, i2 = torch.empty_like(mean), torch.empty_like(istd)
                mom2 = other.momentum if other.momentum is not None else 0.1
                _lib.check(L.eml_bn_finalize_f32(p(sums), C, float(other.eps), float(mom2), p(m2), p(i2),
                                                 p(other.running_mean), p(other.running_var), st), "eml_bn_finalize_f32")
                other.num_batches_tracked += 1
    return mean, istd


class _SpadeNormModulateFn(torch.autograd.Function):
    """``leaky_relu(BN(x) * (1 + gamma) + beta, slope)`` with the parameter-free BatchNorm folded in: the forward
    normalises inline from (mean, istd); the backward emits BatchNorm's reduction (sum dxn, sum dxn*xhat) from the
    modulation pass itself and finishes dx = istd*(dxn - S1/n - xhat*S2/n) in one more streaming pass."""

    @staticmethod
    def forward(ctx, x, gb, mean, istd, slope, training, up2=False):
        """``up2``: ``x`` (B, C, H/2, W/2) is the block input BEFORE the generator's nearest x2 upsample; ``gb`` and the result
        live on the (H, W) grid and the 4x tensor is never written (``eml_spade_norm_modulate_up2_*``)."""
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        _require_gpu_f32(x, "SPADE input")
        B, C = x.shape[:2]
        H, W = gb.shape[2:]
        ctx.up2 = bool(up2)
        if up2:
            x = x.contiguous(memory_format=torch.channels_last)
            gb = gb.contiguous(memory_format=torch.channels_last)
            y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            _lib.check(L.eml_spade_norm_modulate_up2_fwd_f32(p(x), p(gb), p(y), B, H, W, C, float(slope), p(mean), p(istd), st),
                       "eml_spade_norm_modulate_up2_fwd_f32")
        else:
            x, ldx = _rows_view(x)
            gb, ldg = _rows_view(gb)
            y = torch.empty_like(x, memory_format=torch.channels_last)
            _lib.check(L.eml_spade_norm_modulate_fwd_f32(p(x), ldx, p(gb), ldg, p(y), C, B * H * W, C, float(slope), p(mean),
                                                         p(istd), st), "eml_spade_norm_modulate_fwd_f32")
        ctx.save_for_backward(x, gb, mean, istd)
        ctx.slope, ctx.training = float(slope), bool(training)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        x, gb, mean, istd = ctx.saved_tensors
        B, C = x.shape[:2]
        H, W = gb.shape[2:]
        rows = B * H * W
        cl = torch.channels_last
        gy, x, gb = gy.contiguous(memory_format=cl), x.contiguous(memory_format=cl), gb.contiguous(memory_format=cl)
        dgb = torch.empty((B, 2 * C, H, W), dtype=torch.float32, device=x.device, memory_format=cl)
        dxn = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device, memory_format=cl)
        grid = _stats_grid(rows, C)
        # one pass: dxn, dgb, BatchNorm's (sum dxn, sum dxn*xhat) and the column sums of dgb (= the bias gradient of the
        # gamma | beta convolution, handed to its backward on the tensor itself instead of a second read of dgb)
        partials = torch.empty(grid, 4 * C + 1, dtype=torch.float64, device=x.device)
        _lib.check(L.eml_spade_norm_modulate_bwd_cols_f32(p(gy), p(x), p(gb), p(dxn), p(dgb), B, H, W, C, int(ctx.up2), ctx.slope,
                                                          p(mean), p(istd), p(partials), grid, st),
                   "eml_spade_norm_modulate_bwd_cols_f32")
        folded = torch.empty(4 * C + 1, dtype=torch.float64, device=x.device)
        _lib.check(L.eml_bn_fold_f64(p(partials), grid, 4 * C + 1, p(folded), st), "eml_bn_fold_f64")
        sums = None
        if ctx.training:
            sums = folded[:2 * C + 1]
            sums[2 * C:].fill_(float(rows))
            if _bn_sync():
                import torch.distributed as dist
                dist.all_reduce(sums)
        dgb._eml_colsum = (folded[2 * C + 1:].view(C, 2).t().reshape(2 * C).float(), dgb.data_ptr(), dgb._version)
        if ctx.up2:
            dx = torch.empty_like(x, memory_format=cl)     # gradient of the map BEFORE the upsample
            _lib.check(L.eml_bn_bwd_apply_up2_f32(p(dxn), p(x), B, H, W, C, p(mean), p(istd), p(sums), p(dx), st),
                       "eml_bn_bwd_apply_up2_f32")
        else:
            dx = dxn
            _lib.check(L.eml_bn_bwd_apply_f32(p(dx), C, p(x), C, rows, C, p(mean), p(istd), p(sums), p(dx), C, st),
                       "eml_bn_bwd_apply_f32")
        return dx, dgb, None, None, None, None, None


_SPADE_ROWS = {}


def spade_row_order(Cn, device):
    """Row order of the (gamma | beta) weights for ``eml_sphere_conv_spade_fwd_f32`` (include/emlight_hip.h): position p holds
    row ``c + half * Cn`` of cat(gamma head, beta head)."""
    key = (int(Cn), str(device))
    if key not in _SPADE_ROWS:
        pos = torch.arange(2 * Cn)
        c = 64 * (pos // 128) + 32 * ((pos % 128) // 64) + pos % 32
        _SPADE_ROWS[key] = (c + ((pos % 64) // 32) * Cn).to(device)
    return _SPADE_ROWS[key]


def spade_heads_w2(wg, wb, bg, bb, reorder):
    """(W2 (2 Cn, 9 Cin), b2 (2 Cn) or None): SPADE's gamma and beta heads as one operand in the kernels' (tap, c) column order,
    rows in cat order or (``reorder``) in the one-launch SPADE's order -- ONE launch (``eml_spade_heads_w2_f32``) instead of
    cat(weights), cat(biases) and a re-layout copy.  No autograd: callers route the gradients themselves."""
    from .. import _lib
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    Cn, Cin = wg.shape[0], wg.shape[1]
    wg, wb = wg.detach().contiguous(), wb.detach().contiguous()
    has_b = bg is not None and bb is not None
    w2 = torch.empty(2 * Cn, 9 * Cin, dtype=torch.float32, device=wg.device)
    b2 = torch.empty(2 * Cn, dtype=torch.float32, device=wg.device) if has_b else None
    _lib.check(L.eml_spade_heads_w2_f32(p(wg), p(wb), p(bg.detach().contiguous()) if has_b else None,
                                        p(bb.detach().contiguous()) if has_b else None, p(w2), p(b2), Cn, Cin, int(bool(reorder)),
                                        st), "eml_spade_heads_w2_f32")
    return w2, b2


class _SpadeHeadsFn(torch.autograd.Function):
    """cat(gamma head, beta head) of a SPADE as ONE weight (2 Cn, Cin, 3, 3) -- a VIEW of the (O, tap, c) operand the
    SphereConv kernels take without a copy -- and one bias (2 Cn); the gradients are the two halves."""

    @staticmethod
    def forward(ctx, wg, wb, bg, bb):
        Cn, Cin = wg.shape[0], wg.shape[1]
        w2, b2 = spade_heads_w2(wg, wb, bg, bb, False)
        ctx.Cn, ctx.has_b = Cn, b2 is not None
        w = w2.view(2 * Cn, 3, 3, Cin).permute(0, 3, 1, 2)
        if b2 is None:
            b2 = w2.new_zeros(0)
            ctx.mark_non_differentiable(b2)
        return w, b2

    @staticmethod
    def backward(ctx, gw, gb):
        Cn = ctx.Cn
        gwg = gw[:Cn] if gw is not None else None
        gwb = gw[Cn:] if gw is not None else None
        if ctx.has_b and gb is not None:
            return gwg, gwb, gb[:Cn], gb[Cn:]
        return gwg, gwb, None, None


class _SpadeConvModulateFn(torch.autograd.Function):
    """SPADE in one launch: ``leaky_relu(BN(x) * (1 + gamma) + beta, slope)`` with (gamma | beta) = SphereConv(actv; weight,
    bias) formed in the accumulators of the gather-GEMM and consumed by its epilogue -- the (B, 2C, H, W) tensor is neither
    written nor re-read (normalization.py:101-115).  For the backward the forward keeps gamma (C per pixel, half of what the
    two-launch path keeps) and its own output, whose sign is the activation's mask; the (dgamma | dbeta) tensor the modulation's
    backward forms is handed to the SphereConv backward unchanged."""

    @staticmethod
    def forward(ctx, x, actv, wg, wb, bg, bb, mean, istd, slope, training, up2):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        _require_gpu_f32(x, "SPADE input")
        _require_gpu_f32(actv, "SPADE activation")
        B, Cin, H, W = actv.shape
        Cn = x.shape[1]
        want = (B, Cn, H // 2, W // 2) if up2 else (B, Cn, H, W)
        if tuple(x.shape) != want:   # the kernel indexes x from (H, W) alone
            raise ValueError("SPADE: normalised map %s does not match the guide map's grid (expected %s)" % (tuple(x.shape), want))
        cl = torch.channels_last
        geo = sphere_geometry(H, W, 1, actv.device)
        ar = actv.permute(0, 2, 3, 1).contiguous()                 # (B, H, W, Cin); a view when channels-last
        x = x.contiguous(memory_format=cl)
        w2r, br = spade_heads_w2(wg, wb, bg, bb, True)             # rows in the kernel's order: one launch
        y = torch.empty(B * H * W, Cn, dtype=torch.float32, device=x.device)
        keep = any(ctx.needs_input_grad[:6])
        gamma = torch.empty_like(y) if keep else None
        _lib.check(L.eml_sphere_conv_spade_fwd_f32(p(ar), p(geo.idx), p(geo.wgt), p(w2r), p(br) if br is not None else None,
                                                   p(x), p(mean), p(istd), p(y), p(gamma) if keep else None, B, H, W, Cin, Cn,
                                                   int(bool(up2)), float(slope), geo.rowshare, st), "eml_sphere_conv_spade_fwd_f32")
        ctx.geo, ctx.shape, ctx.has_bias = geo, (B, Cin, H, W, 2 * Cn), br is not None
        ctx.up2, ctx.act_slope, ctx.training = bool(up2), float(slope), bool(training)
        # the state _sphere_conv_backward reads: a plain (no epilogue, no kept operand) SphereConv 128 -> 2 Cn
        ctx.slope, ctx.small, ctx.keep = 1.0, False, False
        ctx.fused_dgrad, ctx.fused_wgrad = _backward_dispatch(B, H * W, Cin, 2 * Cn, 1)
        if keep:
            ctx.save_for_backward(ar, wg, wb, x, gamma, y, mean, istd)
        return y.view(B, H, W, Cn).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        ar, wg, wb, x, gamma, y, mean, istd = ctx.saved_tensors
        B, Cin, H, W, O = ctx.shape
        C = O // 2
        rows = B * H * W
        cl = torch.channels_last
        gy = gy.contiguous(memory_format=cl)
        dgb = torch.empty((B, 2 * C, H, W), dtype=torch.float32, device=x.device, memory_format=cl)
        dxn = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device, memory_format=cl)
        grid = _stats_grid(rows, C)
        partials = torch.empty(grid, 4 * C + 1, dtype=torch.float64, device=x.device)
        _lib.check(L.eml_spade_norm_modulate_bwd_y_f32(p(gy), p(x), p(gamma), p(y), p(dxn), p(dgb), B, H, W, C, int(ctx.up2),
                                                       ctx.act_slope, p(mean), p(istd), p(partials), grid, st),
                   "eml_spade_norm_modulate_bwd_y_f32")
        folded = torch.empty(4 * C + 1, dtype=torch.float64, device=x.device)
        _lib.check(L.eml_bn_fold_f64(p(partials), grid, 4 * C + 1, p(folded), st), "eml_bn_fold_f64")
        sums = None
        if ctx.training:
            sums = folded[:2 * C + 1]
            sums[2 * C:].fill_(float(rows))
            if _bn_sync():
                import torch.distributed as dist
                dist.all_reduce(sums)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.up2:
                dx = torch.empty_like(x, memory_format=cl)     # gradient of the map BEFORE the upsample
                _lib.check(L.eml_bn_bwd_apply_up2_f32(p(dxn), p(x), B, H, W, C, p(mean), p(istd), p(sums), p(dx), st),
                           "eml_bn_bwd_apply_up2_f32")
            else:
                dx = dxn
                _lib.check(L.eml_bn_bwd_apply_f32(p(dx), C, p(x), C, rows, C, p(mean), p(istd), p(sums), p(dx), C, st),
                           "eml_bn_bwd_apply_f32")
        # the column sums of dgb = the bias gradient of the gamma | beta convolution (see _SpadeNormModulateFn.backward)
        dgb._eml_colsum = (folded[2 * C + 1:].view(C, 2).t().reshape(2 * C).float(), dgb.data_ptr(), dgb._version)
        nig = ctx.needs_input_grad
        need_w, need_b = nig[2] or nig[3], ctx.has_bias and (nig[4] or nig[5])
        needs = (nig[1], need_w, need_b, False, False, False, False)
        # the input gradient wants the heads in cat order: the same one-launch re-layout, without the row reorder
        wnat = None
        if nig[1]:
            wnat = spade_heads_w2(wg, wb, None, None, False)[0].view(O, 3, 3, Cin).permute(0, 3, 1, 2)
        gactv, gw, gbias = _sphere_conv_backward(ctx, dgb, (ar, wnat), needs)[:3]
        gwg = gw[:C] if (gw is not None and nig[2]) else None
        gwb = gw[C:] if (gw is not None and nig[3]) else None
        gbg = gbias[:C] if (gbias is not None and nig[4]) else None
        gbb = gbias[C:] if (gbias is not None and nig[5]) else None
        return dx, gactv, gwg, gwb, gbg, gbb, None, None, None, None, None


def spade_conv_modulate(x, actv, wg, wb, bg, bb, mean, istd, slope, training, up2):
    """``leaky_relu(((x - mean) * istd) * (1 + gamma) + beta, slope)`` with gamma = SphereConv(actv; wg, bg) and
    beta = SphereConv(actv; wb, bb) as ONE gather-GEMM whose epilogue is the modulation (a module attribute like
    ``sphere_conv``, so that tests can observe the layer shapes that take it)."""
    if not torch.is_grad_enabled():   # a no-grad pass (the discriminator step's generator): nothing is kept for a backward
        wg, wb, bg, bb = wg.detach(), wb.detach(), bg.detach(), bb.detach()
    return _SpadeConvModulateFn.apply(x, actv, wg, wb, bg, bb, mean, istd, slope, training, up2)


def _spade_conv_fusable(x, actv, C, up2):
    """Does this SPADE take the one-launch path?  Where the gamma | beta SphereConv runs on the gather-GEMM kernel anyway, and
    for O = 2C <= 512 also where the library GEMM alone would be a few percent faster: the modulation pass it saves costs more
    (tools/sphere_layers.py: 128 -> 512 at 64 x 128, 2.74 against 2.67 + 0.3 ms)."""
    from .. import _lib
    if not (SphereConv2D.fuse_spade and x.is_cuda and actv.dim() == 4 and actv.shape[0] > 0):
        return False
    B, Cin, H, W = actv.shape
    if up2 and ((H | W) & 1):
        return False
    if not _lib.lib().eml_sphere_conv_spade_supported(Cin, C, H * W):
        return False
    lim = SphereConv2D.fused_min_bytes
    a9_bytes = B * H * W * 9 * Cin * 4
    # (round 5, with the library GEMMs on their recorded selection: sending the 2C = 512 SPADEs down the two-launch path when a
    # backward follows -- so that their weight gradient becomes a library GEMM on the kept operand -- wins 0.36 ms per weight
    # gradient and loses 0.42 ms per forward, the one-launch kernel runs these at 133 TF/s: measured, not adopted)
    return a9_bytes >= 32 * lim or (2 * C <= 256 and a9_bytes >= lim) or (2 * C <= 512 and a9_bytes >= 8 * lim)


def spade_norm_modulate(x, bn, actv, conv_gamma, conv_beta, slope=1.0, stats=None, up2=False):
    """SPADE (normalization.py:101-115) + the LeakyReLU that follows it (architecture.py:56-57; slope 1 = none):
    ``leaky_relu(BN(x) * (1 + gamma(actv)) + beta(actv), slope)``.  gamma | beta come from ONE SphereConv (one gather,
    one GEMM over the concatenated heads); BatchNorm's statistics come from ``spade_batch_stats`` (or ``stats`` when
    the caller already has them for this x) and its normalisation / backward are folded into the modulation kernels."""
    C = x.shape[1]
    wg, wb, bg, bb = conv_gamma.weight, conv_beta.weight, conv_gamma.bias, conv_beta.bias

    def heads():   # cat(gamma head, beta head): one launch on the GPU (the (O, tap, c) operand itself), torch.cat elsewhere
        if wg.is_cuda and wg.dtype == torch.float32 and bg is not None and bb is not None and tuple(wg.shape[2:]) == (3, 3):
            return _SpadeHeadsFn.apply(wg, wb, bg, bb)
        return torch.cat([wg, wb], 0), torch.cat([bg, bb], 0)
    if C % 4 == 0 and isinstance(bn, nn.BatchNorm2d):
        mean, istd = stats if stats is not None else spade_batch_stats(x, bn, repeat=4 if up2 else 1)
        if _spade_conv_fusable(x, actv, C, up2) and bg is not None and bb is not None:
            return spade_conv_modulate(x, actv, wg, wb, bg, bb, mean.detach(), istd.detach(), slope, bn.training, bool(up2))
        w, b = heads()
        gb = sphere_conv(actv, w, b, 1)
        return _SpadeNormModulateFn.apply(x, gb, mean.detach(), istd.detach(), slope, bn.training, bool(up2))
    w, b = heads()
    gb = sphere_conv(actv, w, b, 1)
    if up2:   # ``up2`` = x stands for its nearest x2 upsample (supported by the fused path only; callers check can_fold_up2)
        x = nn.functional.interpolate(x, scale_factor=2)
    # widths that are not a multiple of 4 (never in EMLight) or an instance norm: library norm + elementwise formula
    gamma, beta = torch.split(gb, C, dim=1)
    out = bn(x) * (1 + gamma) + beta
    return out if slope == 1.0 else nn.functional.leaky_relu(out, slope)


spade_norm_modulate.folds_upsample = True   # the oracle's stock-op stand-in does not: callers upsample first


class _SpectralW2Fn(torch.autograd.Function):
    """``weight_orig / sigma`` of ``torch.nn.utils.spectral_norm`` (one power iteration on the ``weight_u`` / ``weight_v``
    buffers in training mode, sigma = u . (W v), u and v constants of the backward) delivered directly in the (O, tap, c)
    memory order of the gather-GEMM kernels: ``csrc/spectral.hip``, 4 launches forward and 2 backward."""

    @staticmethod
    def forward(ctx, w, u, v, iterate, eps, pre=None):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        O, C = w.shape[0], w.shape[1]
        if pre is not None:   # already computed, with every other weight of the network, by spectral_precompute
            w2, uv, sigma = pre
        else:
            w = w.contiguous()
            w2 = torch.empty(O, 9 * C, dtype=torch.float32, device=w.device)
            sigma = torch.empty(1, dtype=torch.float32, device=w.device)
            uv = torch.empty(O + 9 * C, dtype=torch.float32, device=w.device)
            scratch = torch.empty(L.eml_spectral_norm_scratch_floats(O, C), dtype=torch.float32, device=w.device)
            _lib.check(L.eml_spectral_norm_w2_f32(p(w), p(u), p(v), int(bool(iterate)), float(eps), p(w2), p(sigma), p(uv),
                                                  p(scratch), O, C, st), "eml_spectral_norm_w2_f32")
        ctx.save_for_backward(w2, uv, sigma)
        ctx.shape = (O, C)
        ctx.wkey = (id(w), w.data_ptr())   # the parameter this gradient belongs to (its place in a gradient bucket, _dist.grad_slot)
        return w2

    @staticmethod
    def backward(ctx, gw2):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        w2, uv, sigma = ctx.saved_tensors
        O, C = ctx.shape
        gw2 = gw2.contiguous()
        partial = torch.empty(256, dtype=torch.float64, device=gw2.device)
        from .._dist import grad_slot
        dw = grad_slot(ptr=ctx.wkey[1])    # data-parallel runs: straight into the all-reduce bucket (no packing copy)
        if dw is None or tuple(dw.shape) != (O, C, 3, 3) or dw.device != gw2.device:
            dw = torch.empty(O, C, 3, 3, dtype=torch.float32, device=gw2.device)
        _lib.check(L.eml_spectral_norm_w2_bwd_f32(p(gw2), p(w2), p(uv[:O]), p(uv[O:]), p(sigma), p(partial), p(dw), O, C, st),
                   "eml_spectral_norm_w2_bwd_f32")
        return dw, None, None, None, None, None


class _FusedSpectralNormHook:
    """Forward-pre-hook that stands in for torch's ``SpectralNorm`` hook on a SphereConv2D (same parameters and buffers --
    ``weight_orig``, ``weight_u``, ``weight_v`` -- hence the same state_dict): on the GPU the normalised weight comes from
    ``_SpectralW2Fn`` as a (O, C, 3, 3) VIEW of its (O, tap, c) result, which ``_SphereConvFn`` uses without another copy;
    anywhere else (CPU tests on stock ops) torch's own hook runs."""

    def __init__(self, inner):
        self.inner = inner
        self.pre = None   # (key, (w2, uv, sigma)) left by spectral_precompute for this module's NEXT call

    def __getstate__(self):   # torch.save(model) / deepcopy carry the hook, never the GPU scratch of a pending result
        return {"inner": self.inner, "pre": None}

    @staticmethod
    def eligible(sn, w):
        return (w.is_cuda and w.dtype == torch.float32 and sn.dim == 0 and sn.n_power_iterations == 1 and w.dim() == 4
                and tuple(w.shape[2:]) == (3, 3))

    @staticmethod
    def key(module, w, u, v):
        # what a precomputed result was formed from: the parameter and the buffers AS THEY WERE LEFT by the batch (a load_state_dict,
        # an optimizer step or another forward in between changes a version and the result is dropped)
        return (id(w), w._version, id(u), u._version, id(v), v._version, bool(module.training), torch.is_grad_enabled())

    def __call__(self, module, inputs):
        sn = self.inner
        w = getattr(module, sn.name + "_orig")
        if self.eligible(sn, w):
            u, v = getattr(module, sn.name + "_u"), getattr(module, sn.name + "_v")
            O, C = w.shape[0], w.shape[1]
            pre, self.pre = self.pre, None
            if pre is not None and pre[0] != self.key(module, w, u, v):
                pre = None
            w2 = _SpectralW2Fn.apply(w, u, v, module.training, sn.eps, None if pre is None else pre[1])
            wn = w2.view(O, 3, 3, C).permute(0, 3, 1, 2)
            if not isinstance(module, SphereConv2D):
                # an nn.Conv2d (the crop encoder's stride-2 layers) runs on MIOpen, which takes a channels-last WEIGHT down a
                # path 20x slower than its NCHW one (measured: 25.4 against 1.1 ms per joint step, profiles/r05_ab_sn_conv2d.txt):
                # one small copy back to the parameter's own layout
                wn = wn.contiguous()
            setattr(module, sn.name, wn)
        else:
            sn(module, inputs)


# EML_SN_BATCH=0: A/B knob -- every hook launches its own five kernels again
_sn_batch = knob_flag("EML_SN_BATCH", True)
import weakref  # noqa: E402
_SN_PLANS = weakref.WeakKeyDictionary()   # network -> [(module, fused hook)] of spectral_precompute


def spectral_precompute(root):
    """Spectral normalisation of EVERY fused-hook convolution under ``root`` in five launches (``eml_spectral_norm_w2_batch_f32``),
    called at the top of a network's forward: torch's hook -- and round 4's fused one -- run in front of each convolution
    (architecture.py:41-45), 5 launches of 5-10 us each for 23 + 6 weights, twice per iteration.  The power iteration updates
    each module's ``weight_u`` / ``weight_v`` exactly as its own call would (same kernels, same buffers); every hook then
    picks its result up on the module's next call, provided nothing it was formed from has changed (``_FusedSpectralNormHook.key``),
    and falls back to its own launches otherwise -- a module called twice in one forward iterates twice, as the reference's."""
    if not _sn_batch:
        return
    plan = _SN_PLANS.get(root)   # kept beside the module, not in its __dict__: a pickled / copied network does not carry it
    if plan is None:
        plan = [(m, h) for m in root.modules() for h in m._forward_pre_hooks.values() if isinstance(h, _FusedSpectralNormHook)]
        _SN_PLANS[root] = plan
    groups = {}
    for m, h in plan:
        sn = h.inner
        w = getattr(m, sn.name + "_orig", None)
        if w is None or not h.eligible(sn, w) or not w.is_contiguous():
            continue
        groups.setdefault((bool(m.training), float(sn.eps), w.device), []).append((m, h, w))
    if not groups:
        return
    import ctypes
    from .. import _lib
    L, st = _lib.lib(), _lib.current_stream()
    for (iterate, eps, dev), items in groups.items():
        n = len(items)
        if n < 2:
            continue
        shapes = [(w.shape[0], w.shape[1]) for _, _, w in items]
        # one allocation for the small per-weight buffers (sigma | uv | scratch, each 16-byte aligned: the backward reads u and
        # v as float4 where their addresses allow), one per W2 (kept for the backward one by one)
        up4 = lambda k: (k + 3) & ~3
        sizes = [(4, up4(O + 9 * C), up4(L.eml_spectral_norm_scratch_floats(O, C))) for O, C in shapes]
        flat = torch.empty(sum(a + b + c for a, b, c in sizes), dtype=torch.float32, device=dev)
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
        us, vs, w2s, sigmas, uvs, scr, off = [], [], [], [], [], [], 0
        for (m, h, w), (O, C), (a, b, c) in zip(items, shapes, sizes):
            us.append(getattr(m, h.inner.name + "_u"))
            vs.append(getattr(m, h.inner.name + "_v"))
            w2s.append(torch.empty(O, 9 * C, dtype=torch.float32, device=dev))
            sigmas.append(flat[off:off + 1])
            uvs.append(flat[off + a:off + a + O + 9 * C])
            scr.append(flat[off + a + b:off + a + b + c])
            off += a + b + c
        ptrs = lambda ts: arr([t.data_ptr() for t in ts])
        Os, Cs = (ctypes.c_int * n)(*[o for o, _ in shapes]), (ctypes.c_int * n)(*[c for _, c in shapes])
        with torch.no_grad():
            _lib.check(L.eml_spectral_norm_w2_batch_f32(n, ptrs([w for _, _, w in items]), ptrs(us), ptrs(vs), int(iterate), eps,
                                                        ptrs(w2s), ptrs(sigmas), ptrs(uvs), ptrs(scr), Os, Cs, st),
                       "eml_spectral_norm_w2_batch_f32")
        for (m, h, w), u, v, w2, sigma, uv in zip(items, us, vs, w2s, sigmas, uvs):
            h.pre = (h.key(m, w, u, v), (w2, uv, sigma))


def fused_spectral_norm(module):
    """``torch.nn.utils.spectral_norm(module)`` whose per-forward work runs as the HIP kernels above when the weight is a
    3x3 convolution's on the GPU (normalization.py:24-33, architecture.py:41-45)."""
    from torch.nn.utils.spectral_norm import SpectralNorm
    module = torch.nn.utils.spectral_norm(module)
    for key, hook in list(module._forward_pre_hooks.items()):
        if isinstance(hook, SpectralNorm):
            module._forward_pre_hooks[key] = _FusedSpectralNormHook(hook)
    return module


class _InstanceNormActFn(torch.autograd.Function):
    """``leaky_relu(instance_norm(x), slope)`` for ``nn.InstanceNorm2d(affine=False)`` (no running statistics): one HIP launch
    each way (``csrc/instance_norm.hip``) on a channels-last or an NCHW tensor, statistics in f64."""

    @staticmethod
    def forward(ctx, x, eps, slope):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        _require_gpu_f32(x, "InstanceNorm input")
        B, C, H, W = x.shape
        cl = (not x.is_contiguous()) and C % 4 == 0 and x.is_contiguous(memory_format=torch.channels_last)
        if not cl:
            x = x.contiguous()
        y = torch.empty_like(x)      # preserves the layout
        stats = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
        _lib.check(L.eml_instance_norm_act_fwd_f32(p(x), p(y), p(stats), B, H * W, C, int(cl), float(eps), float(slope), st),
                   "eml_instance_norm_act_fwd_f32")
        ctx.save_for_backward(x, stats)
        ctx.cl, ctx.slope = cl, float(slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .. import _lib
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        x, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last) if ctx.cl else gy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(L.eml_instance_norm_act_bwd_f32(p(gy), p(x), p(stats), p(dx), B, H * W, C, int(ctx.cl), ctx.slope, st),
                   "eml_instance_norm_act_bwd_f32")
        return dx, None, None


def instance_norm_act(x, norm, slope=1.0):
    """``leaky_relu(norm(x), slope)`` for a parameter-free ``nn.InstanceNorm2d`` (normalization.py:44-45 + the LeakyReLU that
    follows it in the discriminator / encoder); any other norm module runs as it is, followed by the stock activation."""
    if (isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] * x.shape[3] > 1):
        return _InstanceNormActFn.apply(x, norm.eps, slope)
    y = norm(x)
    return y if slope == 1.0 else nn.functional.leaky_relu(y, slope)


class SphereConv2D(nn.Module):
    """3x3 spherical convolution, same parameters (``weight`` (out,in,3,3), ``bias``) and init as the
    reference (``sphere_cnn.py:87-109``).  Runs on the MI355X only (``sphere_conv`` above); the reference's two
    stock ops (grid_sample + conv2d) are restated in ``oracle/projector.py`` for the tests."""

    keep_operand = True   # unfused layers: keep the im2col operand of a training forward for the weight gradient
    # unit of the fused-kernel thresholds on the size of the im2col operand (see forward); EML_FUSED_MIN_MB: A/B knob
    fused_min_bytes = knob_int("EML_FUSED_MIN_MB", 64, lo=0) << 20
    # SPADE's gamma | beta convolution with the modulation as its epilogue (_SpadeConvModulateFn); EML_FUSE_SPADE=0: A/B knob
    fuse_spade = knob_flag("EML_FUSE_SPADE", True)
    # O <= 4 layers on the one-pass kernels of csrc/sphere_conv_narrow.hip; EML_NARROW=0: A/B knob (im2col + library GEMM)
    narrow_kernels = knob_flag("EML_NARROW", True)
    # round 6: those layers as project-then-gather (csrc/sphere_conv_narrow2.hip); EML_NARROW_PROJECT=0: the one-pass kernels (A/B)
    narrow_project = knob_flag("EML_NARROW_PROJECT", True)
    # low-resolution wide layers on the footprint gather-GEMM (csrc/gather_gemm3.h) instead of im2col + a library GEMM;
    # EML_LOWRES: A/B knob -- off = round 5's dispatch, force = wherever the kernel supports the shape
    lowres = knob_choice("EML_LOWRES", "off", ("auto", "off", "force"))
    # input gradient of the 3-channel input layers through eml_sphere_conv_small_da9_f32; EML_SMALL_DA9=0: A/B knob (general path)
    small_input_grad = knob_flag("EML_SMALL_DA9", True)
    # split-K of the fused weight gradient: workgroups per launch (tiles x K-splits).  Two resident workgroups per CU: 1024 is
    # two full waves of them; round 4's 2048 wrote and re-read twice the partial sums for nothing (same-box A/B of the joint
    # step: 2048 -> 1024 -1.6 ms, 4096 +3 ms; profiles/r05_ab_wgrad_split.txt).  EML_WGRAD_WGS: A/B knob
    wgrad_workgroups = knob_int("EML_WGRAD_WGS", 1024, lo=256)
    # layers whose forward left the im2col operand behind take the library's long-K GEMM for the weight gradient;
    # EML_WGRAD_LIB_KEPT=0: A/B knob (round 4's rule: fused above 256 MB of operand and 32 k pixels)
    lib_wgrad_on_kept_operand = knob_flag("EML_WGRAD_LIB_KEPT", True)

    def __init__(self, in_c, out_c, stride=1, bias=True, mode="bilinear"):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError("SphereConv2D implements the reference's bilinear mode")
        self.in_c, self.out_c, self.stride, self.mode = in_c, out_c, stride, mode
        self.weight = Parameter(torch.empty(out_c, in_c, 3, 3))
        if bias:
            self.bias = Parameter(torch.empty(out_c))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=np.sqrt(5))
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, residual=None, act_slope=1.0):
        if residual is None and act_slope == 1.0:
            return sphere_conv(x, self.weight, self.bias, self.stride)
        return sphere_conv(x, self.weight, self.bias, self.stride, residual, act_slope)
