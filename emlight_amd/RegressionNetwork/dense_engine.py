"""HIP engine of the DenseNet-BC feature extractor (``DenseNet.features`` + relu + avgpool).

Orchestrates the gfx950 kernels of ``csrc/dense_fwd.hip`` / ``csrc/dense_bwd.hip`` through the
C ABI.  PyTorch only owns the memory: one zero-initialised pixel-major (NHWC) buffer per
dense block (the concatenation axis -- ``torch.cat`` of ``DenseNet.py:55`` is a pointer
offset), the 48-channel bottleneck outputs kept for backward, and small per-layer BN
scale/shift / statistics vectors.  With 288 GB of HBM per GPU nothing is recomputed or
freed inside a step; buffers are allocated once per input shape and reused.

Train-mode BatchNorm (the reference never calls ``.eval()`` on this network): each producer
kernel emits f64 partial sums of what it writes, ``eml_dense_bn_prepare_f32`` folds them and
emits the (scale, shift) the next consumer applies while loading its MFMA operand.
"""

import os
import weakref

import numpy as np
import torch

from .. import _lib
from .._knobs import knob_int, knob_flag


def _r16(v):
    return (v + 15) // 16 * 16


def _row_floats(ctot):
    """Row length of a block buffer: whole 128-byte lines (32 floats), so that every pixel row starts on a line.  The
    1x1 kernels read the prefix [0, k) of every row and HBM serves whole lines (FETCH_SIZE per layer,
    profiles/r06_perlayer_1x1.txt): with 304-float rows (block 2) every second row started in the middle of a line and its
    prefix pulled the previous row's unused tail along.  EML_ROW_ALIGN=16: the earlier rounding (A/B)."""
    a = knob_int("EML_ROW_ALIGN", 32, 16, 64)
    if a not in (16, 32, 64):
        raise ValueError("EML_ROW_ALIGN must be 16, 32 or 64")
    return (ctot + a - 1) // a * a


class _Workspace:
    """All device buffers for one (B, H, W, train) shape."""

    def __init__(self, enc, B, H, W, dev, keep_all):
        f32 = dict(dtype=torch.float32, device=dev)
        self.B, self.H, self.W = B, H, W
        self.blocks = []
        h, w = H, W
        for bi, (c0, nl) in enumerate(zip(enc.block_c0, enc.block_layers)):
            ctot = c0 + nl * enc.growth
            blk = {
                "H": h, "W": w, "P": B * h * w, "C0": c0, "Ctot": ctot, "ld": _row_floats(ctot),
                "X": torch.zeros(B * h * w, _row_floats(ctot), **f32),  # zeros: padded / not-yet-written channels stay finite
                "Z": torch.empty(nl if keep_all else 1, B * h * w, enc.inter, **f32),
                "mean": torch.zeros(_r16(ctot), **f32), "var": torch.ones(_r16(ctot), **f32),
                "istd": torch.ones(_r16(ctot), **f32),
                "layers": [],
            }
            for l in range(nl):
                kp = _r16(c0 + l * enc.growth)
                blk["layers"].append({
                    "Cin": c0 + l * enc.growth, "Kp": kp,
                    "scale1": torch.zeros(kp, **f32), "shift1": torch.zeros(kp, **f32),
                    "scale2": torch.zeros(enc.inter, **f32), "shift2": torch.zeros(enc.inter, **f32),
                    "zmean": torch.zeros(enc.inter, **f32), "zvar": torch.ones(enc.inter, **f32),
                    "zistd": torch.ones(enc.inter, **f32),
                    "W1p": torch.empty(kp * 48, **f32), "W2p": torch.empty(9 * 3 * 4 * 16 * 4, **f32),
                    "W2t": torch.empty(7 * 3 * 64 * 4, **f32),   # tap-packed conv3x3 weights (csrc/dense_fwd_tp.hip)
                    # ReLU mask of BN1's output as bits (training only): one 64-bit word per (16-pixel group, 16-channel
                    # K-step, t), whole 256-pixel tiles -- what the data-gradient pass reads instead of X
                    "mask": (torch.empty(((B * h * w + 255) // 256) * 16 * (kp // 16) * 4, dtype=torch.int64, device=dev)
                             if keep_all else None),
                })
            cout = enc.trans_cout[bi]
            kpt = _r16(ctot)
            blk["trans"] = {
                "Cout": cout, "Kp": kpt, "nchunks": (cout + 47) // 48,
                "scale": torch.zeros(kpt, **f32), "shift": torch.zeros(kpt, **f32),
                "Wp": torch.empty(((cout + 47) // 48) * kpt * 48, **f32),
                "Ko": _r16(cout),
                "T": torch.zeros(B * (h // 2) * (w // 2), _r16(cout), **f32),  # raw transition output (kept for backward)
                # pooled activation mean2x2(relu(bn(X))): operand of the transition conv and of its weight gradient
                "A": torch.empty(B * (h // 2) * (w // 2), kpt, **f32),
                # ReLU mask of the transition's BN output, 16 bits per (pooled pixel, channel quad): what the transition's
                # data gradient reads instead of X (training only)
                "mask16": (torch.empty(B * (h // 2) * (w // 2) * (kpt // 4), dtype=torch.int16, device=dev)
                           if keep_all else None),
                "one": torch.ones(kpt, **f32), "zero": torch.zeros(kpt, **f32),
                "tmean": torch.zeros(cout, **f32), "tvar": torch.ones(cout, **f32), "tistd": torch.ones(cout, **f32),
                "scaleL": torch.zeros(cout, **f32), "shiftL": torch.zeros(cout, **f32),
            }
            blk["tp"] = enc.tap_packed_plan(B, h, w, dev)
            self.blocks.append(blk)
            h, w = h // 2, w // 2
        self.hf, self.wf = h, w
        cf = enc.trans_cout[-1]
        self.F = torch.zeros(B * h * w, _r16(cf), **f32)          # last_norm3 output
        self.Y0 = torch.empty(B * H * W, enc.c_init, **f32)       # raw conv0 output (kept for backward)
        self.mean0 = torch.zeros(enc.c_init, **f32)
        self.var0 = torch.ones(enc.c_init, **f32)
        self.istd0 = torch.ones(enc.c_init, **f32)
        self.scale0 = torch.zeros(enc.c_init, **f32)
        self.shift0 = torch.zeros(enc.c_init, **f32)
        # per-workgroup f64 partial sums: forward epilogues write [grid][<= 96 * chunks], the transition dgrad
        # [grid][Kp][2] -- sized from the widest block / transition and the largest grid any launcher may use
        self.partials = torch.zeros(enc.grid_max * max(96 * enc.chunks_max, 2 * enc.kp_max), dtype=torch.float64,
                                    device=dev)
        self.owner, self.done = None, True   # autograd ctx whose backward still needs these buffers (weakref)

    def in_use(self):
        """True while a graph built on this workspace is alive and its backward has not run."""
        return (not self.done) and self.owner is not None and self.owner() is not None


class _PermuteTable:
    """Device array of ``eml_permute_desc`` for ``eml_dense_permute_batch_f32``: every weight re-layout of a pass in ONE
    launch.  Built once per workspace and rebuilt only when a source or destination pointer moved (``model.to()``,
    a re-created parameter); ``load_state_dict`` and the optimiser update weights in place."""
    _DT = np.dtype([("src", "<u8"), ("dst", "<u8"), ("kind", "<i4"), ("Cout", "<i4"), ("Cin", "<i4"), ("Kp", "<i4"),
                    ("Ko", "<i4"), ("reserved", "<i4")])

    def __init__(self):
        self.key, self.dev, self.n = None, None, 0

    def launch(self, L, st, items):
        """items: (weight tensor, destination tensor, kind, Cout, Cin, Kp, Ko)."""
        key = tuple((w.data_ptr(), d.data_ptr(), cout, cin) for w, d, _, cout, cin, *_ in items)
        if key != self.key:
            host = np.zeros(len(items), dtype=self._DT)
            for i, (w, d, kind, cout, cin, kp, ko) in enumerate(items):
                host[i] = (w.data_ptr(), d.data_ptr(), kind, cout, cin, kp, ko, 0)
            self.dev = torch.from_numpy(host.view(np.uint8).copy()).to(items[0][1].device)
            self.key, self.n = key, len(items)
        _lib.check(L.eml_dense_permute_batch_f32(_lib.ptr(self.dev), self.n, st), "eml_dense_permute_batch_f32")


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, x, *params):
        ctx.enc = enc
        needs_grad = any(ctx.needs_input_grad[2:])
        ws, pooled = enc.run_forward(x, keep_all=needs_grad)
        ctx.ws = ws
        if needs_grad:   # the buffers now belong to this graph until its backward ran (or the graph is dropped)
            ws.owner, ws.done = weakref.ref(ctx), False
        ctx.save_for_backward(x)
        return pooled

    @staticmethod
    def backward(ctx, gpooled):
        (x,) = ctx.saved_tensors
        ws = ctx.ws
        if ws.owner is None or ws.owner() is not ctx:
            raise RuntimeError("HIP DenseNet: the activations of this forward are gone -- backward() was already run on "
                               "this graph (the workspace is released after the first backward; retain_graph / double "
                               "backward are not supported)")
        grads = ctx.enc.run_backward(ws, x, gpooled.contiguous())
        # the buffers go back to the pool: a second backward on this graph (retain_graph=True) raises above instead of
        # reading activations a later forward may have overwritten
        ws.done, ws.owner = True, None
        ctx.enc.trim_pool(ws)
        return (None, None) + tuple(grads)


class HipDenseEncoder:
    GRID_MAX = 1024   # upper bound of every persistent grid (scratch buffers are sized for it)

    @staticmethod
    def check_supported(model):
        """The kernels are built for EMLight's DenseNet-BC (growth 12, bottleneck 48); per-channel vectors live in
        384-entry LDS / coefficient arrays.  Reject everything else up front instead of corrupting memory."""
        f = model.features
        if model.bn_size * model.growth_rate != 48 or model.growth_rate != 12:
            raise NotImplementedError("the HIP engine is built for growth_rate=12, bn_size=4 (EMLight's DenseNet)")
        c = f.conv0.out_channels
        if c != 24:   # conv0_fwd_kernel / conv0_bwd_weight_kernel are instantiated for EMLight's 24 only
            raise NotImplementedError("the HIP engine is built for num_init_features=24 (got %d)" % c)
        for bi, nl in enumerate(model.block_config):
            ctot = c + nl * model.growth_rate
            if nl < 1 or _r16(ctot) > 384:
                raise NotImplementedError("dense block %d would be %d channels wide; the HIP kernels hold per-channel "
                                          "vectors of at most 384 (EMLight: 216/300/342)" % (bi + 1, ctot))
            c = getattr(f, "transition%d" % (bi + 1)).conv.out_channels
            if _r16(c) == 48:   # the 1x1 launchers tell a dense layer (48 outputs) from a transition by that width
                raise NotImplementedError("transition %d has %d output channels: widths that pad to 48 collide with the "
                                          "dense layers' bottleneck in the 1x1 launchers (EMLight: 108/150/171)" % (bi + 1, c))
            if c % 2 and bi + 1 < len(model.block_config):   # the next block's channel offsets must stay 8-byte aligned
                raise NotImplementedError("transition %d feeds a dense block with an odd channel count (%d)" % (bi + 1, c))

    def __init__(self, model):
        f = model.features
        self.check_supported(model)
        self.model = model
        self.growth = model.growth_rate
        self.inter = model.bn_size * model.growth_rate
        self.c_init = f.conv0.out_channels
        self.block_layers = list(model.block_config)
        self.block_c0, self.trans_cout = [], []
        c = self.c_init
        for bi, nl in enumerate(self.block_layers):
            self.block_c0.append(c)
            c += nl * self.growth
            c = getattr(f, "transition%d" % (bi + 1)).conv.out_channels
            self.trans_cout.append(c)
        self.avgpool = model.avgpool_size
        self.grid_max = self.GRID_MAX
        self.kp_max = max(_r16(c0 + nl * self.growth) for c0, nl in zip(self.block_c0, self.block_layers))
        self.ko_max = max(_r16(c) for c in self.trans_cout)
        self.chunks_max = max((c + 47) // 48 for c in self.trans_cout)
        self._ws = {}
        self._cu = None

    # ------------------------------------------------------------------ parameter plumbing
    def param_list(self):
        f = self.model.features
        ps = [f.conv0.weight, f.norm0.weight, f.norm0.bias]
        for bi, nl in enumerate(self.block_layers):
            blk = getattr(f, "denseblock%d" % (bi + 1))
            for l in range(nl):
                L = getattr(blk, "denselayer%d" % (l + 1))
                ps += [L.norm1.weight, L.norm1.bias, L.conv1.weight, L.norm2.weight, L.norm2.bias, L.conv2.weight]
            T = getattr(f, "transition%d" % (bi + 1))
            ps += [T.norm.weight, T.norm.bias, T.conv.weight]
            LN = getattr(f, "last_norm%d" % (bi + 1))
            ps += [LN.weight, LN.bias]
        return ps

    def __call__(self, x):
        x = _lib.require_gpu_tensor(x, "x")
        B, C, H, W = x.shape
        if C != 3:
            raise ValueError("expected (B,3,H,W) input")
        div = 2 ** len(self.block_layers)
        if H % div or W % div or (H // div) < self.avgpool or (W // div) < self.avgpool:
            raise ValueError("crop %dx%d: each of the %d transitions halves the map and the HIP kernels need even maps "
                             "(H, W divisible by %d), then the head pools %dx%d.  Deviation from the reference: its "
                             "AvgPool2d(2) floors odd maps, so e.g. 100x132 trains there; crop to a multiple of %d "
                             "(EMLight's 192x256 and 240x320 are)." % (H, W, len(self.block_layers), div, self.avgpool,
                                                                       self.avgpool, div))
        return _EncoderFn.apply(self, x, *self.param_list())

    def _grid(self, dev):
        if self._cu is None:
            self._cu = torch.cuda.get_device_properties(dev).multi_processor_count
        return self._tuned("EML_GRID", 2 * self._cu)

    def _tuned(self, env, default):
        """Persistent-grid size; the environment variable is a tuning knob for A/B runs (multiple of 8)."""
        g = int(os.environ.get(env, 0))
        return min(self.grid_max, g if g > 0 else default)

    def overlap_wgrad(self, dev):
        """Run the conv3x3 weight gradients (MFMA-bound, off the critical path) on a side stream next to the HBM-bound
        1x1 chain of the backward.  EML_WGRAD_OVERLAP=0/1 overrides the default."""
        return torch.device(dev).type == "cuda" and os.environ.get("EML_WGRAD_OVERLAP", "0") == "1"

    def _grid3(self, dev):
        """Persistent grid of the conv3x3 kernels: their 512-thread workgroups use ~140 KB of LDS, one fits a CU, and a
        second round of workgroups only repeats the weight-fragment prologue (A/B on one box: #CU beats 2 x #CU by 4 % on
        the three kernels, 1.8 ms per step; the HBM-bound 1x1 kernels do not care between 2x and 4x #CU)."""
        self._grid(dev)
        return self._tuned("EML_GRID3", self._cu)

    def tap_packed_plan(self, B, H, W, dev):
        """(band_rows, grid) when the conv3x3 forward of a block of (B, H, W) runs the tap-packed kernel
        (csrc/dense_fwd_tp.hip: 84 MFMAs per 16 pixels instead of 108, no halo tile), else None.  Measured per geometry
        (tools/bench_c3tp.py, profiles/r06_c3tp_*): it wins where four wavefronts side by side cover the image width
        (W = 320 or 256: block 1, 0.81 -> 0.62 ms per layer at B = 64), ties at two (block 2) and loses at one (block 3),
        where the bands needed to fill the chip get too short for their halo rows.  Train-mode BatchNorm only (the mode the
        reference runs this network in, train.py:42 / test.py:36-37): with untrained running statistics the eval-mode
        logits are O(100) and f32 round-off is 1e-4 there -- the reference's own f32 result is 8.8e-5 from the f64 one
        (profiles/r06_c3tp_golden_err.txt) -- so eval mode keeps the summation order its golden outputs were pinned with.
        EML_C3_TP=off: the halo-tile kernel everywhere (A/B)."""
        from .._knobs import knob_choice
        if knob_choice("EML_C3_TP", "auto", ("auto", "off")) == "off":
            return None
        if _lib.lib().eml_dense_conv3x3_fwd_tp_supported(B, H, W) != 4 or H < 8:
            return None
        self._grid(dev)
        bands = max(1, min(H // 4, -(-2 * self._cu // B)))       # ~2 workgroups per CU, bands of >= 4 rows
        band = -(-H // bands)
        return band, min(self.grid_max, B * (-(-H // band)))

    def workspace(self, B, H, W, dev, keep_all):
        """Buffers for this shape.  A workspace still owned by a live graph (a second grad-enabled forward of the same
        shape before the first backward: summed losses, gradient accumulation, GAN-style double forward) is never
        handed out again -- that forward gets its own buffers."""
        key = (B, H, W, dev, keep_all)
        pool = self._ws.get(key)
        if pool is None:
            pool = []
            self._ws = {key: pool}  # one live shape at a time (buffers are GBs at training sizes)
        for ws in pool:
            if not ws.in_use():
                return ws
        if pool:
            import warnings
            warnings.warn("HIP DenseNet: a %s grad-enabled forward of shape %s is alive before the previous one ran its "
                          "backward: allocating another full activation workspace (GBs at training sizes; idle extras are "
                          "freed after a backward)" % ("second" if len(pool) == 1 else "%d-th" % (len(pool) + 1),
                                                       (B, H, W)), stacklevel=3)
        ws = _Workspace(self, B, H, W, dev, keep_all)
        pool.append(ws)
        return ws

    POOL_KEEP = 2   # idle workspaces (with their backward buffers) kept per shape after a backward

    def trim_pool(self, ws):
        """After a backward: drop idle workspaces beyond POOL_KEEP so a one-off double forward (or a burst of accumulated,
        un-backwarded losses) does not leave N full activation sets resident for the rest of training."""
        for key, pool in self._ws.items():
            if ws in pool:
                idle = [w for w in pool if not w.in_use()]
                for extra in idle[self.POOL_KEEP:]:
                    pool.remove(extra)
                return

    # ------------------------------------------------------------------ forward
    def _prepare(self, L, st, partials, G, pstride, n_new, c_new0, count, mean, var, istd, bn, C, Cpad, training,
                 scale, shift):
        p = _lib.ptr
        if bn is None:
            args = (None, None, None, None)
            eps, mom = 1e-5, 0.1
        else:
            args = (p(bn.weight), p(bn.bias), p(bn.running_mean), p(bn.running_var))
            eps, mom = bn.eps, (bn.momentum if bn.momentum is not None else 0.1)
        _lib.check(L.eml_dense_bn_prepare_f32(p(partials), G, pstride, n_new, c_new0, float(count), p(mean), p(var),
                                              p(istd), *args, C, Cpad, eps, mom, int(training), p(scale), p(shift),
                                              st), "eml_dense_bn_prepare_f32")

    def run_forward(self, x, keep_all):
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
        m = self.model
        f = m.features
        B, _, H, W = x.shape
        dev = x.device
        ws = self.workspace(B, H, W, dev, keep_all)
        G = self._grid(dev)
        G3 = self._grid3(dev)
        Gf = self._tuned("EML_GRID_FWD1", G)   # per-family knob for A/B runs (default: the common 2 x #CU)
        Gb = min(self.grid_max, 4 * (self._cu or 256))
        training = m.training
        part = ws.partials
        b0 = ws.blocks[0]
        # ---- every weight re-layout of the forward (1x1 -> W1p / Wp, 3x3 -> W2p) in one launch
        if getattr(ws, "fwd_permutes", None) is None:
            ws.fwd_permutes = _PermuteTable()
        items = []
        for bi, blk in enumerate(ws.blocks):
            mod = getattr(f, "denseblock%d" % (bi + 1))
            for l, lay in enumerate(blk["layers"]):
                Lm = getattr(mod, "denselayer%d" % (l + 1))
                items.append((Lm.conv1.weight, lay["W1p"], 0, 48, lay["Cin"], lay["Kp"], 0))
                if blk["tp"] is None or not training:
                    items.append((Lm.conv2.weight, lay["W2p"], 1, 12, 48, 48, 0))
                else:
                    items.append((Lm.conv2.weight, lay["W2t"], 3, 12, 48, 48, 0))
            T, tr = getattr(f, "transition%d" % (bi + 1)), blk["trans"]
            items.append((T.conv.weight, tr["Wp"], 0, tr["Cout"], blk["Ctot"], tr["Kp"], 0))
        ws.fwd_permutes.launch(L, st, items)
        # ---- conv0 -> norm0 -> relu0 (DenseNet.py:88-93)
        # conv0 on the matrix unit (round 6): outputs and BatchNorm partial sums bit for bit those of the VALU kernel (the MFMA adds
        # its 27 terms in the fma chain's order; the sums are formed in that kernel's association), -0.35 ms per step;
        # EML_CONV0_MFMA=0: the VALU kernel (A/B)
        conv0 = ("eml_dense_conv0_fwd_mfma_f32" if knob_flag("EML_CONV0_MFMA", True) and 3 * B * H * W < 2 ** 31   # (32-bit offsets)
                 else "eml_dense_conv0_fwd_f32")
        _lib.check(getattr(L, conv0)(p(x), p(f.conv0.weight), p(ws.Y0), self.c_init, B, H, W, self.c_init,
                                     p(part), G, st), conv0)
        self._prepare(L, st, part, G, 2 * self.c_init, self.c_init, 0, b0["P"], ws.mean0, ws.var0, ws.istd0, f.norm0,
                      self.c_init, self.c_init, training, ws.scale0, ws.shift0)
        _lib.check(L.eml_dense_bn_apply_f32(p(ws.Y0), self.c_init, p(b0["X"]), b0["ld"], self.c_init, b0["P"],
                                            p(ws.scale0), p(ws.shift0), 1, p(part), Gb, st), "eml_dense_bn_apply_f32")
        self._prepare(L, st, part, Gb, 2 * self.c_init, self.c_init, 0, b0["P"], b0["mean"], b0["var"], b0["istd"],
                      None, 0, 0, training, None, None)
        nb = len(ws.blocks)
        for bi, blk in enumerate(ws.blocks):
            mod = getattr(f, "denseblock%d" % (bi + 1))
            P, Hb, Wb, ld = blk["P"], blk["H"], blk["W"], blk["ld"]
            pending = False  # partial stats of the previous layer's 12 new channels wait in `part`
            Gc = G3          # ... in that many rows (the grid of the conv3x3 kernel that wrote them)
            for l, lay in enumerate(blk["layers"]):
                Lm = getattr(mod, "denselayer%d" % (l + 1))
                cin, kp = lay["Cin"], lay["Kp"]
                z = blk["Z"][l if keep_all else 0]
                self._prepare(L, st, part if pending else None, Gc, 32, 12, cin - 12, P, blk["mean"], blk["var"],
                              blk["istd"], Lm.norm1, cin, kp, training, lay["scale1"], lay["shift1"])
                _lib.check(L.eml_dense_conv1x1_fwd_f32(p(blk["X"]), ld, P, Hb, Wb, 0, kp, p(lay["scale1"]),
                                                       p(lay["shift1"]), p(lay["W1p"]), 48, p(z), 48, p(part), Gf,
                                                       p(lay["mask"]) if (keep_all and training) else None, st),
                           "eml_dense_conv1x1_fwd_f32")
                self._prepare(L, st, part, Gf, 96, 48, 0, P, lay["zmean"], lay["zvar"], lay["zistd"], Lm.norm2, 48, 48,
                              training, lay["scale2"], lay["shift2"])
                if blk["tp"] is None or not training:
                    Gc = G3
                    _lib.check(L.eml_dense_conv3x3_fwd_f32(p(z), p(lay["scale2"]), p(lay["shift2"]), p(lay["W2p"]),
                                                           p(blk["X"]), ld, cin, B, Hb, Wb, p(part), G3, st),
                               "eml_dense_conv3x3_fwd_f32")
                else:
                    band, Gc = blk["tp"]
                    _lib.check(L.eml_dense_conv3x3_fwd_tp_f32(p(z), p(lay["scale2"]), p(lay["shift2"]), p(lay["W2t"]),
                                                              p(blk["X"]), ld, cin, B, Hb, Wb, band, p(part), Gc, st),
                               "eml_dense_conv3x3_fwd_tp_f32")
                pending = True
            # ---- transition (BN-ReLU-1x1-avgpool2, DenseNet.py:14-21) + last_norm (DenseNet.py:122)
            tr, T = blk["trans"], getattr(f, "transition%d" % (bi + 1))
            LN = getattr(f, "last_norm%d" % (bi + 1))
            ctot, cout, kpt = blk["Ctot"], tr["Cout"], tr["Kp"]
            self._prepare(L, st, part if pending else None, Gc, 32, 12, ctot - 12, P, blk["mean"], blk["var"],
                          blk["istd"], T.norm, ctot, kpt, training, tr["scale"], tr["shift"])
            Pn = B * (Hb // 2) * (Wb // 2)
            # pool first (it commutes with the 1x1 conv): the conv's output chunks then read A, a quarter of X
            _lib.check(L.eml_dense_pool_act_f32(p(blk["X"]), ld, B, Hb, Wb, kpt, p(tr["scale"]), p(tr["shift"]),
                                                p(tr["A"]), kpt, p(tr["mask16"]) if (keep_all and training) else None,
                                                st), "eml_dense_pool_act_f32")
            _lib.check(L.eml_dense_conv1x1_fwd_f32(p(tr["A"]), kpt, Pn, Hb // 2, Wb // 2, 0, kpt, p(tr["one"]),
                                                   p(tr["zero"]), p(tr["Wp"]), cout, p(tr["T"]), tr["Ko"], p(part), G,
                                                   None, st), "eml_dense_conv1x1_fwd_f32(transition)")
            for ch in range(tr["nchunks"]):
                nv = min(48, cout - 48 * ch)
                last = ch == tr["nchunks"] - 1
                pv = part[ch * G * 96:]
                self._prepare(L, st, pv, G, 96, nv, 48 * ch, Pn, tr["tmean"], tr["tvar"], tr["tistd"],
                              LN if last else None, cout if last else 0, cout if last else 0, training,
                              tr["scaleL"] if last else None, tr["shiftL"] if last else None)
            if bi + 1 < nb:
                nxt = ws.blocks[bi + 1]
                dst, ldd = nxt["X"], nxt["ld"]
            else:
                dst, ldd = ws.F, ws.F.shape[1]
            _lib.check(L.eml_dense_bn_apply_f32(p(tr["T"]), tr["Ko"], p(dst), ldd, cout, Pn, p(tr["scaleL"]),
                                                p(tr["shiftL"]), 0, p(part), Gb, st), "eml_dense_bn_apply_f32")
            if bi + 1 < nb:
                self._prepare(L, st, part, Gb, 2 * cout, cout, 0, Pn, nxt["mean"], nxt["var"], nxt["istd"], None, 0, 0,
                              training, None, None)
        # ---- relu -> avgpool(k) -> flatten in (C,h,w) order (DenseNet.py:136-137)
        cf = self.trans_cout[-1]
        k = self.avgpool
        pooled = torch.empty(B, cf * (ws.hf // k) * (ws.wf // k), dtype=torch.float32, device=dev)
        _lib.check(L.eml_dense_head_pool_fwd_f32(p(ws.F), ws.F.shape[1], cf, B, ws.hf, ws.wf, k, p(pooled), st),
                   "eml_dense_head_pool_fwd_f32")
        if training:
            bns = [mm for mm in f.modules() if isinstance(mm, torch.nn.BatchNorm2d)]
            torch._foreach_add_([mm.num_batches_tracked for mm in bns], 1)
        return ws, pooled

    # ------------------------------------------------------------------ backward
    def run_backward(self, ws, x, gpooled):
        from .dense_engine_bwd import run_backward
        return run_backward(self, ws, x, gpooled)
