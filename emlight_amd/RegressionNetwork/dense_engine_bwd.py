"""Backward pass of the HIP DenseNet-BC engine: orchestration of ``csrc/dense_bwd.hip``.

Walks the network in reverse (head -> transition3/last_norm3 -> block3 -> ... -> conv0),
keeping one gradient buffer G per dense block that mirrors the block's NHWC activation
buffer; every BatchNorm backward is reduced to a per-channel affine (dx = cA*dy + cB*x + cC)
and folded into the operand loads of the neighbouring convolution kernels.
"""
import ctypes
import os

import torch

from .. import _lib
from .._knobs import knob_choice, knob_flag


def _r16(v):
    return (v + 15) // 16 * 16


def _side_stream(dev, prio):
    """A second HIP stream on `dev`.  torch's pool only hands out priorities <= 0; a LOW-priority stream (prio > 0: the
    background weight gradients must not take CUs from the critical chain) is created through the HIP runtime torch
    has already loaded and wrapped as an ExternalStream."""
    if prio <= 0:
        return torch.cuda.Stream(device=dev, priority=prio)
    hip = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(prio))  # 1 = non-blocking
    if rc != 0:
        raise RuntimeError("hipStreamCreateWithPriority failed (%d)" % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


def _masked_stream(dev, n_cu, keep):
    """A HIP stream restricted to the CUs i of [0, n_cu) with keep(i) (hipExtStreamCreateWithCUMask): a kernel launched
    on it only ever occupies that subset, so a persistent grid sized for the subset never waits for a CU that another
    stream's kernel holds."""
    hip = ctypes.CDLL("libamdhip64.so")
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cu):
        if keep(i):
            mask[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


class _BwdBuffers:
    def __init__(self, enc, ws, dev):
        f32 = dict(dtype=torch.float32, device=dev)
        self.G = [torch.zeros(blk["P"], blk["ld"], **f32) for blk in ws.blocks]
        self.GF = torch.zeros(ws.F.shape[0], ws.F.shape[1], **f32)
        maxP = ws.blocks[0]["P"]
        self.DZ = torch.empty(2, maxP, 48, **f32)   # one per layer of a pair
        # finished gradient of a layer's 12 output channels (compact); a ring, so that the conv3x3 weight gradient of
        # layer l may still be reading slot l % R on the side stream while the main stream is R - 1 layers further
        self.ring = max(1, int(os.environ.get("EML_WGRAD_RING", 4))) if enc.overlap_wgrad(dev) else 1
        self.GF12 = [torch.empty(maxP, 12, **f32) for _ in range(self.ring)]
        self.N12 = torch.empty(maxP, 12, **f32)     # narrow pass: finished gradient of the lower layer's 12 channels
        # the two-layer data-gradient pass's top 24 channels (the next pair's output channels) as two compact tensors: their
        # readers then fetch 48 bytes per pixel instead of one or two 128-byte lines of a wide G row (EML_DGRAD_TOP=0: A/B)
        self.TOP = torch.empty(2 * maxP * 12, **f32) if knob_flag("EML_DGRAD_TOP", True) else None   # (2, P, 12) per block
        # scratch sized from the network (widest block Kp, widest transition Ko) and the largest grid, not for
        # EMLight's default only
        kp, ko, g = enc.kp_max, max(enc.ko_max, 48), enc.grid_max
        # permuted conv1 / transition weights for the data gradient: one buffer per layer (all of them are re-laid out by
        # ONE launch at the start of the backward)
        self.Wd = [[torch.empty(lay["Kp"] * 48, **f32) for lay in blk["layers"]] for blk in ws.blocks]
        self.WdT = [torch.empty(blk["trans"]["Kp"] * blk["trans"]["Ko"], **f32) for blk in ws.blocks]
        self.permutes = None
        self.coef = torch.zeros(8, max(kp, ko), **f32)  # two (cA,cB,cC) sets for dz, then sB, sC
        self.part2 = torch.zeros(2, g * kp * 2, dtype=torch.float64, device=dev)
        # weight-gradient partials: conv1x1 [grid][Kp][48], conv3x3 [2*grid][27*256], conv0 [4*grid][1024]
        self.partW = torch.empty(g * max(kp * 48, 2 * 27 * 256, 4 * 1024), **f32)
        # BN1 / transition-norm dgamma: channels whose weight-gradient identity is ill-conditioned (small |gamma|) are flagged
        # by the finalize kernel and recomputed directly (eml_dense_bn_dgamma_direct_f32; gated on the device by the
        # flags of the SAME backward -- no host-side report).
        i32 = dict(dtype=torch.int32, device=dev)
        self.cond = [[torch.zeros(lay["Kp"], **i32) for lay in blk["layers"]] for blk in ws.blocks]
        self.condT = [torch.zeros(blk["trans"]["Kp"], **i32) for blk in ws.blocks]
        self.any_ill = torch.zeros(1, **i32)
        self.dg_grid = 128
        self.dg_scratch = torch.empty(self.dg_grid * kp, dtype=torch.float64, device=dev)
        # side stream of the conv3x3 weight gradients (nothing downstream waits for dW2): own partial buffers per slot
        self.side = None
        if enc.overlap_wgrad(dev):
            # EML_CU_SPLIT=m: the side stream owns the CUs with i % m == m - 1, the main chain of the backward runs on a
            # stream that owns the rest (experiment: partitioned instead of competing for the same CUs)
            m = int(os.environ.get("EML_CU_SPLIT", 0))
            self.main_masked = None
            if m > 1:
                n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
                self.side = _masked_stream(dev, n_cu, lambda i: i % m == m - 1)
                self.main_masked = _masked_stream(dev, n_cu, lambda i: i % m != m - 1)
            else:
                self.side = _side_stream(dev, int(os.environ.get("EML_SIDE_PRIO", 0)))
            self.partW3 = [torch.empty(g * 2 * 27 * 256, **f32) for _ in range(self.ring)]
            self.ev_ready = [torch.cuda.Event() for _ in range(self.ring)]
            self.ev_done = [None] * self.ring


def run_backward(enc, ws, x, gpooled):
    if getattr(ws, "bwd", None) is None:
        ws.bwd = _BwdBuffers(enc, ws, x.device)
    masked = getattr(ws.bwd, "main_masked", None)
    if masked is None:
        return _run_backward(enc, ws, x, gpooled)
    cur = torch.cuda.current_stream(x.device)
    masked.wait_stream(cur)
    with torch.cuda.stream(masked):
        out = _run_backward(enc, ws, x, gpooled)
    cur.wait_stream(masked)
    return out


def _run_backward(enc, ws, x, gpooled):
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    m = enc.model
    f = m.features
    if not m.training:
        raise NotImplementedError("HIP DenseNet backward is implemented for train-mode BatchNorm "
                                  "(EMLight trains and tests in train mode, train.py:42 / test.py:36-37)")
    dev = x.device
    B, _, H, W = x.shape
    G = enc._grid(dev)
    G3 = enc._grid3(dev)   # conv3x3 kernels: one 512-thread workgroup per CU (see HipDenseEncoder._grid3)
    # (EML_D3_SHORT=1: the data gradient's 256-thread / 4-row-tile A/B geometry wants two workgroups per CU, csrc/dense_bwd.hip)
    G3d = enc._tuned("EML_GRID3_DGRAD", 2 * G3 if knob_flag("EML_D3_SHORT", False) else G3)
    Gw = enc._tuned("EML_GRID_WGRAD1", G)   # per-family knobs for A/B runs (default: the common 2 x #CU)
    Gd = enc._tuned("EML_GRID_DGRAD", G)
    Gb = min(enc.grid_max, 4 * enc._cu)
    bw = ws.bwd
    part = ws.partials
    ring = [0]
    if bw.side is not None:
        main = torch.cuda.current_stream(dev)
        st_side = ctypes.c_void_p(bw.side.cuda_stream)
        G3s = enc._tuned("EML_GRID3_SIDE", G3)
        bw.side.wait_stream(main)   # the parameter-gradient tensors below are allocated on the main stream
    # ---- every data-gradient weight layout of the backward in one launch
    if bw.permutes is None:
        from .dense_engine import _PermuteTable
        bw.permutes = _PermuteTable()
    items = []
    for bi_, blk_ in enumerate(ws.blocks):
        mod_ = getattr(f, "denseblock%d" % (bi_ + 1))
        for l_, lay_ in enumerate(blk_["layers"]):
            items.append((getattr(mod_, "denselayer%d" % (l_ + 1)).conv1.weight, bw.Wd[bi_][l_], 2, 48, lay_["Cin"],
                          lay_["Kp"], 48))
        tr_ = blk_["trans"]
        items.append((getattr(f, "transition%d" % (bi_ + 1)).conv.weight, bw.WdT[bi_], 2, tr_["Cout"], blk_["Ctot"],
                      tr_["Kp"], tr_["Ko"]))
    bw.permutes.launch(L, st, items)
    params = enc.param_list()
    from .._dist import grad_slot
    grads = {}
    for q in params:   # data-parallel runs: the kernels write straight into the all-reduce bucket (no packing copy)
        slot = grad_slot(q)
        grads[id(q)] = slot if slot is not None else torch.empty_like(q)
    gr = lambda q: p(grads[id(q)])
    coefs = [tuple(bw.coef[3 * k + i] for i in range(3)) for k in range(2)]
    cA, cB, cC = coefs[0]
    sB, sC = bw.coef[6], bw.coef[7]

    # ---- ill-conditioned dgamma channels (|gamma| < 1e-3 |beta|, flagged by THIS backward's finalize kernels on the
    # device): the direct recomputation is always enqueued and returns at once when its layer flagged nothing.  (Round 3
    # skipped the launches while the newest COMPLETED report of an earlier backward said "none flagged"; that report lags
    # the parameters by a step or more, so a channel that had just crossed the threshold kept the noisy quotient -- or 0
    # for gamma == 0 -- for those steps: ADVICE round 3.  The ~100 empty launches cost 0.3 % of the step.)
    # EML_DGAMMA_DIRECT=never exists for A/B timing only.
    direct = knob_choice("EML_DGAMMA_DIRECT", "always", ("always", "never")) != "never"
    c3_fold = knob_flag("EML_C3_FOLD", True)   # read once, not per layer
    bw.any_ill.zero_()

    def dgamma_direct(blk, X_ld, Pin, Hin, Win, pool, DY, ld_dy, Zr, ld_z, co, Cout, conv, Cin, scale1, shift1, cond, bn):
        if not direct:
            return
        a, b, c = co if co is not None else (None, None, None)
        _lib.check(L.eml_dense_bn_dgamma_direct_f32(
            p(blk["X"]), X_ld, Pin, Hin, Win, pool, p(DY), ld_dy, p(Zr), ld_z, p(a), p(b), p(c), Cout, p(conv.weight), Cin,
            p(scale1), p(shift1), p(blk["mean"]), p(blk["istd"]), p(cond), p(bw.dg_scratch), gr(bn.weight), bw.dg_grid, st),
            "eml_dense_bn_dgamma_direct_f32")

    def finalize(R, pstride, count, bn, mean, istd, C, Cpad, coef=0, s_acc=None, src=None, c_lo=0, c_hi=None, conv=None,
                 cond=None):
        """dgamma/dbeta of `bn` for channels [c_lo, c_hi); coef: write the dz affine into coefficient set
        `coef` (None: skip); s_acc: fold (cB,cC) into sB/sC (True: add, False: overwrite).
        conv: the 1x1 conv that follows bn + ReLU -- S2 then comes from its finished weight gradient
        (sum_o W*dW = sum dy*bn(x)); the masked data-gradient passes accumulate S1 only."""
        a, b, c = coefs[coef] if coef is not None else (None, None, None)
        wargs = (p(bn.bias), p(conv.weight), gr(conv.weight), conv.weight.shape[0]) if conv is not None else (None, None, None, 0)
        _lib.check(L.eml_dense_bn_bwd_finalize_f32(
            p(part if src is None else src), R, pstride, float(count), p(bn.weight), p(mean), p(istd), C, Cpad, 1,
            gr(bn.weight), gr(bn.bias), p(a), p(b), p(c),
            p(sB) if s_acc is not None else None, p(sC) if s_acc is not None else None, int(bool(s_acc)),
            c_lo, Cpad if c_hi is None else c_hi, *wargs, p(cond), p(bw.any_ill) if cond is not None else None, st),
            "eml_dense_bn_bwd_finalize_f32")

    def parr(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def materialize(Gbuf, blk, c0, n):
        _lib.check(L.eml_dense_grad_materialize_f32(p(Gbuf), blk["ld"], p(blk["X"]), blk["ld"], p(sB), p(sC), c0, n,
                                                    blk["P"], st), "eml_dense_grad_materialize_f32")

    # ---- head: relu -> avgpool(k) backward
    cf, k = enc.trans_cout[-1], enc.avgpool
    ldF = ws.F.shape[1]
    _lib.check(L.eml_dense_head_pool_bwd_f32(p(gpooled), p(ws.F), ldF, cf, B, ws.hf, ws.wf, k, p(bw.GF), ldF, st),
               "eml_dense_head_pool_bwd_f32")
    dY, ld_dy = bw.GF, ldF
    nb = len(ws.blocks)
    for bi in reversed(range(nb)):
        blk, tr = ws.blocks[bi], ws.blocks[bi]["trans"]
        mod = getattr(f, "denseblock%d" % (bi + 1))
        T, LN = getattr(f, "transition%d" % (bi + 1)), getattr(f, "last_norm%d" % (bi + 1))
        P, Hb, Wb, ld, ctot = blk["P"], blk["H"], blk["W"], blk["ld"], blk["Ctot"]
        cout, Ko, kpt = tr["Cout"], tr["Ko"], tr["Kp"]
        Pn = B * (Hb // 2) * (Wb // 2)
        Gbuf = bw.G[bi]
        top2 = bw.TOP[:2 * P * 12].view(2, P, 12) if bw.TOP is not None else None
        # ---- last_norm backward (affine folded into the transition kernels' dz operand)
        _lib.check(L.eml_dense_bn_bwd_stats_f32(p(dY), ld_dy, p(tr["T"]), Ko, None, 0, 0, cout, Pn, p(tr["tmean"]),
                                                p(tr["tistd"]), p(part), Gb, st), "eml_dense_bn_bwd_stats_f32")
        finalize(Gb, 2 * cout, Pn, LN, tr["tmean"], tr["tistd"], cout, Ko, coef=0)
        # ---- transition conv (pool folded): weight grad, data grad, BN backward -> G (write)
        # weight grad on the pooled activation A kept by the forward (unit BN: relu(1*A + 0) == A)
        _lib.check(L.eml_dense_conv1x1_bwd_weight_f32(
            p(tr["A"]), kpt, Pn, Hb // 2, Wb // 2, 0, kpt, ctot, p(tr["one"]), p(tr["zero"]), p(dY), ld_dy, p(tr["T"]),
            Ko, p(cA), p(cB), p(cC), cout, p(bw.partW), gr(T.conv.weight), G, None, None, 0, None, 0, None, None, st),
            "eml_dense_conv1x1_bwd_weight_f32")
        _lib.check(L.eml_dense_conv1x1_bwd_data_f32(
            p(dY), ld_dy, p(tr["T"]), Ko, p(cA), p(cB), p(cC), Ko, p(bw.WdT[bi]), None, ld, p(tr["scale"]),
            p(tr["shift"]), None, None, Pn, Hb, Wb, 1, kpt, p(Gbuf), ld, 0, p(part), G, p(tr["mask16"]), st),
            "eml_dense_conv1x1_bwd_data_f32")   # ReLU mask from pool_act's bits: X is not read
        finalize(G, 2 * kpt, P, T.norm, blk["mean"], blk["istd"], ctot, kpt, coef=None, s_acc=False, conv=T.conv,
                 cond=bw.condT[bi])
        dgamma_direct(blk, ld, P, Hb, Wb, 1, dY, ld_dy, tr["T"], Ko, coefs[0], cout, T.conv, ctot, tr["scale"], tr["shift"],
                      bw.condT[bi], T.norm)   # before the dense layers reuse coefficient set 0
        # ---- dense layers, last to first, two per pass of the block gradient (see dense_bwd.hip:
        #      "dense layers: 1 or 2 layers per pass")
        def conv2_backward(l, slot, n12=False, narrow_lo=None, top=False):
            """conv3x3 backward of layer l -> DZ[slot], dW2, BN2 backward -> coefficient set `slot`; conv1 wgrad, which
            also materialises dz = cA*dzn + cB*z + cC in place over DZ[slot] for the data-gradient passes.
            n12: the layer above left the finished gradient of this layer's channels in the compact bw.N12.
            narrow_lo: this is the upper layer of a pair -- its narrow data pass over the lower layer's 12 output channels
            [narrow_lo, narrow_lo + 12) rides on the weight-gradient kernel's dz tile (-> bw.N12, S1 -> bw.part2[slot]).
            top: the pair above left this pair's gradient columns in the compact bw.TOP (its data-gradient pass wrote them
            there instead of into G): [1] = this (upper) layer's 12 channels, [0] = the lower layer's (the narrow operand)."""
            lay = blk["layers"][l]
            Lm = getattr(mod, "denselayer%d" % (l + 1))
            cin, kp, z, dz = lay["Cin"], lay["Kp"], blk["Z"][l], bw.DZ[slot]
            # the gradient of this layer's 12 output channels is complete: its deferred BN1 affine (sB, sC) is
            # applied inside the conv3x3 dgrad's tile staging, which also leaves the finished gradient in GF12
            gsrc = (p(bw.N12), 12, 0) if n12 else (p(top2[1]), 12, 0) if top else (p(Gbuf), ld, cin)
            r = ring[0] = (ring[0] + 1) % bw.ring
            if bw.side is not None and bw.ev_done[r] is not None:
                main.wait_event(bw.ev_done[r])   # the side stream's weight gradient that last read this slot
            # round 4: data gradient + weight gradient of the layer in one pass over the tiles (16-byte staging loads in
            # blocks 1 and 2, pairs of 8-byte ones in block 3, which starts at channel 150) -- EML_C3_FOLD=0: the two launches (A/B)
            fold = (bw.side is None and c3_fold
                    and L.eml_dense_conv3x3_bwd_fused_supported(gsrc[1], gsrc[2], ld, cin) == 1)
            if fold:
                _lib.check(L.eml_dense_conv3x3_bwd_fused_f32(*gsrc, p(Lm.conv2.weight), p(z), p(lay["zmean"]), p(lay["zistd"]),
                                                             p(dz), B, Hb, Wb, p(part), G3d, p(blk["X"]), ld, cin, p(sB),
                                                             p(sC), p(bw.GF12[r]), p(lay["scale2"]), p(lay["shift2"]),
                                                             p(bw.partW), gr(Lm.conv2.weight), st),
                           "eml_dense_conv3x3_bwd_fused_f32")
            else:
                _lib.check(L.eml_dense_conv3x3_bwd_data_f32(*gsrc, p(Lm.conv2.weight), p(z), p(lay["zmean"]),
                                                            p(lay["zistd"]), p(dz), B, Hb, Wb, p(part), G3d, p(blk["X"]), ld,
                                                            cin, p(sB), p(sC), p(bw.GF12[r]), st),
                           "eml_dense_conv3x3_bwd_data_f32")
            if fold:
                pass
            elif bw.side is None:
                _lib.check(L.eml_dense_conv3x3_bwd_weight_f32(p(bw.GF12[r]), 12, 0, p(z), p(lay["scale2"]),
                                                              p(lay["shift2"]), B, Hb, Wb, p(bw.partW),
                                                              gr(Lm.conv2.weight), G3, st),
                           "eml_dense_conv3x3_bwd_weight_f32")
            else:
                # dW2 feeds nothing downstream: MFMA-bound, nearly HBM-idle -> side stream, next to the HBM-bound 1x1 chain
                bw.ev_ready[r].record(main)
                bw.side.wait_event(bw.ev_ready[r])
                _lib.check(L.eml_dense_conv3x3_bwd_weight_f32(p(bw.GF12[r]), 12, 0, p(z), p(lay["scale2"]),
                                                              p(lay["shift2"]), B, Hb, Wb, p(bw.partW3[r]),
                                                              gr(Lm.conv2.weight), G3s, st_side),
                           "eml_dense_conv3x3_bwd_weight_f32")
                if bw.ev_done[r] is None:
                    bw.ev_done[r] = torch.cuda.Event()
                bw.ev_done[r].record(bw.side)
            finalize(G3d, 96, P, Lm.norm2, lay["zmean"], lay["zistd"], 48, 48, coef=slot)
            a, b, c = coefs[slot]
            _lib.check(L.eml_dense_conv1x1_bwd_weight_f32(
                p(blk["X"]), ld, P, Hb, Wb, 0, kp, cin, p(lay["scale1"]), p(lay["shift1"]), p(dz), 48, p(z), 48,
                p(a), p(b), p(c), 48, p(bw.partW), gr(Lm.conv1.weight), Gw, p(dz),
                *((p(Lm.conv1.weight), narrow_lo, *((p(top2[0]), 12) if top else (p(Gbuf), ld)), p(bw.N12),
                   p(bw.part2[slot])) if narrow_lo is not None
                  else (None, 0, None, 0, None, None)), st), "eml_dense_conv1x1_bwd_weight_f32")
            return Lm

        def dgrad(layers, slots, k_lo, k_hi, top=False):
            """G[:, k_lo:k_hi] += sum over `layers` of scale1*dam; BN1 partial sums -> bw.part2[slot].
            top: columns [k_hi - 24, k_hi) -- the next pair's output channels -- go to bw.TOP instead of G."""
            lays = [blk["layers"][l] for l in layers]
            if top:
                _lib.check(L.eml_dense_conv1x1_bwd_data_multi_top_f32(
                    parr([bw.DZ[s_] for s_ in slots]), parr([bw.Wd[bi][l_] for l_ in layers]),
                    parr([y["scale1"] for y in lays]), parr([y["shift1"] for y in lays]),
                    parr([bw.part2[s_] for s_ in slots]), (ctypes.c_int * len(layers))(*[y["Kp"] for y in lays]),
                    P, k_hi, p(Gbuf), ld, Gd, parr([y["mask"] for y in lays]), p(top2), st),
                    "eml_dense_conv1x1_bwd_data_multi_top_f32")
                return
            _lib.check(L.eml_dense_conv1x1_bwd_data_multi_f32(
                len(layers), parr([bw.DZ[s_] for s_ in slots]), None, None, None, None,   # DZ holds the materialised dz
                parr([bw.Wd[bi][l_] for l_ in layers]),
                parr([y["scale1"] for y in lays]), parr([y["shift1"] for y in lays]),
                parr([bw.part2[s_] for s_ in slots]), (ctypes.c_int * len(layers))(*[y["Kp"] for y in lays]),
                None, ld, None, None, P, k_lo, k_hi, p(Gbuf), ld, Gd,
                parr([y["mask"] for y in lays]), st),   # ReLU masks from the forward's bits: X is not read
                "eml_dense_conv1x1_bwd_data_multi_f32")

        def bn1_finalize(l, Lm, slot, c_lo, c_hi, rows=None):
            """rows: the grid of the kernel that wrote this channel range's partials (wide pass: Gd; narrow pass: Gw)"""
            lay = blk["layers"][l]
            finalize(Gd if rows is None else rows, 2 * lay["Kp"], P, Lm.norm1, blk["mean"], blk["istd"], lay["Cin"], lay["Kp"], coef=None,
                     s_acc=True, src=bw.part2[slot], c_lo=c_lo, c_hi=c_hi, conv=Lm.conv1, cond=bw.cond[bi][l])

        def bn1_direct(l, Lm, slot):
            """after the layer's last bn1_finalize, while DZ[slot] still holds its materialised dz"""
            lay = blk["layers"][l]
            dgamma_direct(blk, ld, P, Hb, Wb, 0, bw.DZ[slot], 48, None, 0, None, 48, Lm.conv1, lay["Cin"], lay["scale1"],
                          lay["shift1"], bw.cond[bi][l], Lm.norm1)

        l = len(blk["layers"]) - 1
        have_top = False   # the pair above left this pair's 24 gradient columns in top2
        while l >= 0:
            if l >= 1:
                la, lb = l, l - 1
                cin_a, cin_b = blk["layers"][la]["Cin"], blk["layers"][lb]["Cin"]
                Lma = conv2_backward(la, 0, narrow_lo=cin_b, top=have_top)   # + narrow pass: layer lb's output channels only -> N12
                bn1_finalize(la, Lma, 0, cin_b, cin_a, rows=Gw)
                Lmb = conv2_backward(lb, 1, n12=True)
                # the next pair (lb - 1, lb - 2) owns columns [cin_b - 24, cin_b): masked passes only (the bits the forward
                # kept), and only where the fused conv3x3 backward / the narrow epilogue are what reads them
                to_top = (top2 is not None and lb >= 2 and blk["layers"][la]["mask"] is not None and bw.side is None
                          and cin_b >= 24 and cin_b % 4 == 0)   # (block 3 starts at channel 150: quads straddle)
                dgrad([la, lb], [0, 1], 0, cin_b, top=to_top)  # both layers, X read once, G updated once
                have_top = to_top
                bn1_finalize(la, Lma, 0, 0, cin_b)
                bn1_finalize(lb, Lmb, 1, 0, blk["layers"][lb]["Kp"])
                bn1_direct(la, Lma, 0)
                bn1_direct(lb, Lmb, 1)
                l -= 2
            else:
                Lm0 = conv2_backward(l, 0)
                dgrad([l], [0], 0, blk["layers"][l]["Cin"])
                bn1_finalize(l, Lm0, 0, 0, blk["layers"][l]["Kp"])
                bn1_direct(l, Lm0, 0)
                l -= 1
        c0b = blk["C0"]
        # block input channels: every layer has contributed.  Block 1's are norm0 + relu0's output: their deferred affine is
        # applied inside the two kernels that read them (round 6: no pass over G, no read of the block buffer's first line)
        norm0_fused = bi == 0 and knob_flag("EML_NORM0_FUSED", True) and c0b % 4 == 0 and c0b <= 32
        if not norm0_fused:
            materialize(Gbuf, blk, 0, c0b)
        dY, ld_dy = Gbuf, ld
    # ---- relu0 / norm0 / conv0
    b0 = ws.blocks[0]
    c0 = enc.c_init
    if norm0_fused:
        _lib.check(L.eml_dense_norm0_bwd_stats_f32(p(dY), ld_dy, p(ws.Y0), c0, p(ws.scale0), p(ws.shift0), p(sB), p(sC), c0,
                                                   b0["P"], p(ws.mean0), p(ws.istd0), p(part), Gb, st),
                   "eml_dense_norm0_bwd_stats_f32")
    else:
        _lib.check(L.eml_dense_bn_bwd_stats_f32(p(dY), ld_dy, p(ws.Y0), c0, p(b0["X"]), b0["ld"], 1, c0, b0["P"],
                                                p(ws.mean0), p(ws.istd0), p(part), Gb, st), "eml_dense_bn_bwd_stats_f32")
    finalize(Gb, 2 * c0, b0["P"], f.norm0, ws.mean0, ws.istd0, c0, _r16(c0), coef=0)
    if norm0_fused:
        _lib.check(L.eml_dense_conv0_bwd_weight_fused_f32(p(x), p(dY), ld_dy, p(ws.Y0), c0, p(ws.scale0), p(ws.shift0), p(sB),
                                                          p(sC), p(cA), p(cB), p(cC), B, H, W, p(bw.partW),
                                                          gr(f.conv0.weight), G, st), "eml_dense_conv0_bwd_weight_fused_f32")
    else:
        _lib.check(L.eml_dense_conv0_bwd_weight_f32(p(x), p(dY), ld_dy, p(b0["X"]), b0["ld"], p(ws.Y0), c0, p(cA), p(cB),
                                                    p(cC), B, H, W, p(bw.partW), gr(f.conv0.weight), G, st),
                   "eml_dense_conv0_bwd_weight_f32")
    if bw.side is not None:
        main.wait_stream(bw.side)   # every dW2 is complete before the gradients leave
    return [grads[id(q)] for q in params]
