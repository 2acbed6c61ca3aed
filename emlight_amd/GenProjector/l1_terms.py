"""The generator's L1-type loss terms on the MI355X: mask-weighted feature matching (``models/pix2pix_model.py:99-117``) and
the VGG perceptual term (``models/networks/loss.py:102-114``), all pairs of a term in ONE launch each way
(``csrc/losses.hip``, ``eml_l1_pairs_{fwd,bwd}_f32``) instead of ~12 ATen launches and as many passes over the feature maps
per pair.  There is no CPU path here: the CPU tests run the stock formulas (``pix2pix_model.py`` picks per device)."""
import ctypes

import torch

from .. import _lib
from .._knobs import knob_flag

ENABLED = knob_flag("EML_FUSED_L1", True)   # EML_FUSED_L1=0: A/B knob (the ATen formulas on the GPU too)


def _rows(t):
    """(B, C, H, W) tensor -> (its channels-last dense version, pixel rows, C)."""
    t = t.contiguous(memory_format=torch.channels_last)   # a no-op for the HIP kernels' outputs
    b, c, h, w = t.shape
    return t, b * h * w, c


class _L1PairsFn(torch.autograd.Function):
    """``sum_i scale_i * sum |f_i - r_i| |w_i|`` -> (1,) tensor.  ``spec``: per pair (kind, scale) with kind "halves" (f, r = the
    two batch halves of ONE tensor -- a discriminator feature map of cat(fake, real): its gradient spans both halves, the real
    half zero) or "pair" (two tensors; gradient for the first only).  ``tensors``: per pair the map(s), then the per-pixel
    weight map (B, 1, H, W) or None -- flattened in pair order: halves -> (t, w), pair -> (f, r, w)."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        L, st = _lib.lib(), _lib.current_stream()
        n = len(spec)
        f, r, w, rows, C, scale, keep, meta = [], [], [], [], [], [], [], []
        it = iter(tensors)
        for kind, s in spec:
            if kind == "halves":
                t, wm = next(it), next(it)
                t, nrow, c = _rows(t)
                if t.shape[0] % 2:
                    raise ValueError("feature map of cat(fake, real) with an odd batch")
                half = t.numel() // 2
                f.append(t.data_ptr())
                r.append(t.data_ptr() + 4 * half)
                rows.append(nrow // 2)
                meta.append(("halves", t.shape, half))
                keep.append(t)
            else:
                a, b, wm = next(it), next(it), next(it)
                a, nrow, c = _rows(a)
                b, nrow_b, c_b = _rows(b)
                if (nrow, c) != (nrow_b, c_b):
                    raise ValueError("L1 pair of different shapes: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
                f.append(a.data_ptr())
                r.append(b.data_ptr())
                rows.append(nrow)
                meta.append(("pair", a.shape, 0))
                keep += [a, b]
            if wm is not None:
                wm = wm.contiguous()
                if wm.numel() != rows[-1] or wm.dtype != torch.float32:
                    raise ValueError("per-pixel weight map %s does not match %d pixel rows" % (tuple(wm.shape), rows[-1]))
                keep.append(wm)
            w.append(wm.data_ptr() if wm is not None else None)
            C.append(c)
            scale.append(float(s) / float(rows[-1] * c) if rows[-1] else 0.0)
        for t in keep:
            if not t.is_cuda or t.dtype != torch.float32:
                raise _lib.EmlightHipError("the fused L1 terms take float32 tensors on the MI355X; there is no CPU path")
        dev = keep[0].device
        arrs = ((ctypes.c_void_p * n)(*f), (ctypes.c_void_p * n)(*r), (ctypes.c_void_p * n)(*w), (ctypes.c_long * n)(*rows),
                (ctypes.c_int * n)(*C), (ctypes.c_float * n)(*scale))
        partial = torch.empty(L.eml_l1_pairs_partial_doubles(n), dtype=torch.float64, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(L.eml_l1_pairs_fwd_f32(n, *arrs, _lib.ptr(partial), _lib.ptr(out), st), "eml_l1_pairs_fwd_f32")
        ctx.arrs, ctx.meta, ctx.n = arrs, meta, n
        ctx.spec = spec
        ctx.save_for_backward(*keep)   # keeps the storages behind the recorded pointers alive
        return out

    @staticmethod
    def backward(ctx, gout):
        L, st = _lib.lib(), _lib.current_stream()
        _ = ctx.saved_tensors
        n = ctx.n
        gout = gout.contiguous().float()
        g, gz, nz, grads = [], [], [], []
        for (kind, shape, half) in ctx.meta:
            gt = torch.empty(shape, dtype=torch.float32, device=gout.device, memory_format=torch.channels_last)
            g.append(gt.data_ptr())
            if kind == "halves":
                gz.append(gt.data_ptr() + 4 * half)
                nz.append(half)
                grads += [gt, None]
            else:
                gz.append(None)
                nz.append(0)
                grads += [gt, None, None]
        _lib.check(L.eml_l1_pairs_bwd_f32(n, *ctx.arrs, _lib.ptr(gout), (ctypes.c_void_p * n)(*g), (ctypes.c_void_p * n)(*gz),
                                          (ctypes.c_long * n)(*nz), st), "eml_l1_pairs_bwd_f32")
        return (None, *grads)


MAX_PAIRS = 16   # pairs per launch: the kernels take their pair table by value in the kernel arguments (csrc/losses.hip)


def l1_pairs(spec, tensors):
    """See ``_L1PairsFn``; returns a 0-dim tensor.  Any number of pairs: a launch takes up to ``MAX_PAIRS`` (the reference's
    defaults make 8 feature-matching and 5 VGG terms; ``--num_D 3 --n_layers_D 6`` makes 18), more run as further launches
    whose results are summed; none (``--n_layers_D 1``: no intermediate maps) is the reference's zero loss."""
    spec, tensors = tuple(spec), list(tensors)
    if not spec:
        like = next((t for t in tensors if isinstance(t, torch.Tensor)), None)
        return torch.zeros((), dtype=torch.float32, device=like.device if like is not None else "cuda")
    width = [2 if kind == "halves" else 3 for kind, _ in spec]
    total, k, off = None, 0, 0
    while k < len(spec):
        n = min(MAX_PAIRS, len(spec) - k)
        m = sum(width[k:k + n])
        part = _L1PairsFn.apply(spec[k:k + n], *tensors[off:off + m]).reshape(())
        total = part if total is None else total + part
        k, off = k + n, off + m
    return total


def feature_matching(feats, masks, num_D):
    """``sum_ij mean(|(f_ij - r_ij) * (50 - 49 m_ij)|) / num_D`` over the discriminators' intermediate outputs, given as the
    maps of cat(fake, real) (``feats``: flat list) with the light mask at each map's resolution (``masks``: (B, 1, h, w))."""
    spec, tensors = [], []
    for t, m in zip(feats, masks):
        spec.append(("halves", 1.0 / num_D))
        tensors += [t, 50.0 - 49.0 * m]
    return l1_pairs(spec, tensors)


def l1_sum(pairs):
    """``sum_i w_i * L1(a_i, b_i)`` (mean reduction; gradient to a_i only): ``pairs`` = iterable of (w_i, a_i, b_i)."""
    spec, tensors = [], []
    for wgt, a, b in pairs:
        spec.append(("pair", float(wgt)))
        tensors += [a, b.detach(), None]
    return l1_pairs(spec, tensors)
