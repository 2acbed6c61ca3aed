"""Backward pass of the HIP DenseNet-BC engine: orchestration of ``csrc/dense_bwd.hip``.

Walks the network in reverse (head -> transition3/last_norm3 -> block3 -> ... -> conv0),
keeping one gradient buffer G per dense block that mirrors the block's NHWC activation
buffer; every BatchNorm backward is reduced to a per-channel affine (dx = cA*dy + cB*x + cC)
and folded into the operand loads of the neighbouring convolution kernels.
"""
import torch

from .. import _lib


def _r16(v):
    return (v + 15) // 16 * 16


class _BwdBuffers:
    def __init__(self, enc, ws, dev):
        f32 = dict(dtype=torch.float32, device=dev)
        self.G = [torch.zeros(blk["P"], blk["ld"], **f32) for blk in ws.blocks]
        self.GF = torch.zeros(ws.F.shape[0], ws.F.shape[1], **f32)
        maxP = ws.blocks[0]["P"]
        self.DZ = torch.empty(maxP, 48, **f32)
        self.Wd = torch.empty(352 * 176, **f32)
        self.coef = torch.zeros(6, 384, **f32)  # cA,cB,cC for the "dz" side and for the "dx" side
        g = enc.grid_max
        self.partW = torch.empty(max(g * 2 * 352 * 48 // 2, g * 27 * 256, g * 4 * 1024), **f32)


def run_backward(enc, ws, x, gpooled):
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
    m = enc.model
    f = m.features
    if not m.training:
        raise NotImplementedError("HIP DenseNet backward is implemented for train-mode BatchNorm "
                                  "(EMLight trains and tests in train mode, train.py:42 / test.py:36-37)")
    dev = x.device
    B, _, H, W = x.shape
    G = enc._grid(dev)
    Gb = min(enc.grid_max, 4 * enc._cu)
    if getattr(ws, "bwd", None) is None:
        ws.bwd = _BwdBuffers(enc, ws, dev)
    bw = ws.bwd
    part = ws.partials
    params = enc.param_list()
    grads = {id(q): torch.empty_like(q) for q in params}
    gr = lambda q: p(grads[id(q)])
    cA, cB, cC, sB, sC = (bw.coef[i] for i in range(5))

    def finalize(R, pstride, count, bn, mean, istd, C, Cpad, coef=True, s_acc=None):
        """dgamma/dbeta of `bn`; coef: write the dz affine (cA,cB,cC); s_acc: fold (cB,cC) into sB/sC."""
        _lib.check(L.eml_dense_bn_bwd_finalize_f32(
            p(part), R, pstride, float(count), p(bn.weight), p(mean), p(istd), C, Cpad, 1, gr(bn.weight), gr(bn.bias),
            p(cA) if coef else None, p(cB) if coef else None, p(cC) if coef else None,
            p(sB) if s_acc is not None else None, p(sC) if s_acc is not None else None, int(bool(s_acc)), st),
            "eml_dense_bn_bwd_finalize_f32")

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
        # ---- last_norm backward (affine folded into the transition kernels' dz operand)
        _lib.check(L.eml_dense_bn_bwd_stats_f32(p(dY), ld_dy, p(tr["T"]), Ko, None, 0, 0, cout, Pn, p(tr["tmean"]),
                                                p(tr["tistd"]), p(part), Gb, st), "eml_dense_bn_bwd_stats_f32")
        finalize(Gb, 2 * cout, Pn, LN, tr["tmean"], tr["tistd"], cout, Ko)
        # ---- transition conv (pool folded): weight grad, data grad, BN backward -> G (write)
        _lib.check(L.eml_dense_conv1x1_bwd_weight_f32(
            p(blk["X"]), ld, Pn, Hb, Wb, 1, kpt, ctot, p(tr["scale"]), p(tr["shift"]), p(dY), ld_dy, p(tr["T"]), Ko,
            p(cA), p(cB), p(cC), cout, p(bw.partW), gr(T.conv.weight), G, st), "eml_dense_conv1x1_bwd_weight_f32")
        _lib.check(L.eml_dense_permute_w1_bwd_f32(p(T.conv.weight), cout, ctot, kpt, Ko, p(bw.Wd), st),
                   "eml_dense_permute_w1_bwd_f32")
        _lib.check(L.eml_dense_conv1x1_bwd_data_f32(
            p(dY), ld_dy, p(tr["T"]), Ko, p(cA), p(cB), p(cC), Ko, p(bw.Wd), p(blk["X"]), ld, p(tr["scale"]),
            p(tr["shift"]), p(blk["mean"]), p(blk["istd"]), Pn, Hb, Wb, 1, kpt, p(Gbuf), ld, 0, p(part), G, st),
            "eml_dense_conv1x1_bwd_data_f32")
        finalize(G, 2 * kpt, P, T.norm, blk["mean"], blk["istd"], ctot, kpt, coef=False, s_acc=False)
        # ---- dense layers, last to first
        for l in reversed(range(len(blk["layers"]))):
            lay = blk["layers"][l]
            Lm = getattr(mod, "denselayer%d" % (l + 1))
            cin, kp = lay["Cin"], lay["Kp"]
            z = blk["Z"][l]
            materialize(Gbuf, blk, cin, 12)  # gradient of this layer's 12 output channels is complete
            _lib.check(L.eml_dense_conv3x3_bwd_data_f32(p(Gbuf), ld, cin, p(Lm.conv2.weight), p(z), p(lay["zmean"]),
                                                        p(lay["zistd"]), p(bw.DZ), B, Hb, Wb, p(part), G, st),
                       "eml_dense_conv3x3_bwd_data_f32")
            _lib.check(L.eml_dense_conv3x3_bwd_weight_f32(p(Gbuf), ld, cin, p(z), p(lay["scale2"]), p(lay["shift2"]),
                                                          B, Hb, Wb, p(bw.partW), gr(Lm.conv2.weight), G, st),
                       "eml_dense_conv3x3_bwd_weight_f32")
            finalize(G, 96, P, Lm.norm2, lay["zmean"], lay["zistd"], 48, 48)
            _lib.check(L.eml_dense_conv1x1_bwd_weight_f32(
                p(blk["X"]), ld, P, Hb, Wb, 0, kp, cin, p(lay["scale1"]), p(lay["shift1"]), p(bw.DZ), 48, p(z), 48,
                p(cA), p(cB), p(cC), 48, p(bw.partW), gr(Lm.conv1.weight), G, st), "eml_dense_conv1x1_bwd_weight_f32")
            _lib.check(L.eml_dense_permute_w1_bwd_f32(p(Lm.conv1.weight), 48, cin, kp, 48, p(bw.Wd), st),
                       "eml_dense_permute_w1_bwd_f32")
            _lib.check(L.eml_dense_conv1x1_bwd_data_f32(
                p(bw.DZ), 48, p(z), 48, p(cA), p(cB), p(cC), 48, p(bw.Wd), p(blk["X"]), ld, p(lay["scale1"]),
                p(lay["shift1"]), p(blk["mean"]), p(blk["istd"]), P, Hb, Wb, 0, kp, p(Gbuf), ld, 1, p(part), G, st),
                "eml_dense_conv1x1_bwd_data_f32")
            finalize(G, 2 * kp, P, Lm.norm1, blk["mean"], blk["istd"], cin, kp, coef=False, s_acc=True)
        c0b = blk["C0"]
        materialize(Gbuf, blk, 0, c0b)  # block input channels: every layer has contributed
        dY, ld_dy = Gbuf, ld
    # ---- relu0 / norm0 / conv0
    b0 = ws.blocks[0]
    c0 = enc.c_init
    _lib.check(L.eml_dense_bn_bwd_stats_f32(p(dY), ld_dy, p(ws.Y0), c0, p(b0["X"]), b0["ld"], 1, c0, b0["P"],
                                            p(ws.mean0), p(ws.istd0), p(part), Gb, st), "eml_dense_bn_bwd_stats_f32")
    finalize(Gb, 2 * c0, b0["P"], f.norm0, ws.mean0, ws.istd0, c0, _r16(c0))
    _lib.check(L.eml_dense_conv0_bwd_weight_f32(p(x), p(dY), ld_dy, p(b0["X"]), b0["ld"], p(ws.Y0), c0, p(cA), p(cB),
                                                p(cC), B, H, W, p(bw.partW), gr(f.conv0.weight), G, st),
               "eml_dense_conv0_bwd_weight_f32")
    return [grads[id(q)] for q in params]
