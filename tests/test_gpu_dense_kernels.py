"""GPU: every DenseNet-engine kernel of libemlight_hip.so, one launcher at a time, against the same op
written in plain PyTorch (f64 on the GPU) on identical inputs.  Shapes are deliberately ragged (pixel
counts that are not multiples of the 256-pixel tiles, H/W that are not multiples of the 8x32 conv
tile, channel counts that are not multiples of 16) so every bounds path runs.  Tolerances are f32
round-off for the reduction length involved."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = 64  # persistent workgroups used by the tests (the engine uses 2 x #CU)


@pytest.fixture(scope="module")
def lib():
    from emlight_amd import _lib
    return _lib


def r16(v):
    return (v + 15) // 16 * 16


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, device=DEV) * scale


def close(got, want, rtol=2e-5, atol=None, what=""):
    want = want.double()
    atol = atol if atol is not None else 2e-5 * float(want.abs().max() + 1e-30)
    err = (got.double() - want).abs()
    assert bool((err <= atol + rtol * want.abs()).all()), "%s: max err %.3e (scale %.3e)" % (
        what, float(err.max()), float(want.abs().max()))


def fold_partials(part, rows, nch):
    """[rows][nch][2] f64 partials -> (sum, sumsq)"""
    v = part[:rows * nch * 2].view(rows, nch, 2).sum(0)
    return v[:, 0], v[:, 1]


def nhwc(t):  # (B,C,H,W) -> (B*H*W, C)
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def nchw(t, B, H, W):  # (P, C) -> (B,C,H,W)
    return t.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("entry,B,H,W,ld", [("eml_dense_conv0_fwd_f32", 2, 20, 44, 32), ("eml_dense_conv0_fwd_mfma_f32", 2, 20, 44, 32),
                                            ("eml_dense_conv0_fwd_mfma_f32", 3, 17, 23, 24), ("eml_dense_conv0_fwd_mfma_f32", 1, 5, 3, 24)])
def test_conv0_fwd_and_stats(lib, entry, B, H, W, ld):
    """Both kernels of the layer (VALU: eval mode; matrix unit: train mode; ragged sizes: the last tile is partial, rows shorter
    than a 16-pixel group) against F.conv2d in f64, with the BatchNorm sums of what was written."""
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    x, w0 = torch.rand(B, 3, H, W, device=DEV), rnd(24, 3, 3, 3, scale=0.3)
    Y = torch.zeros(B * H * W, ld, device=DEV)
    part = torch.zeros(G * 48, dtype=torch.float64, device=DEV)
    lib.check(getattr(L, entry)(p(x), p(w0), p(Y), ld, B, H, W, 24, p(part), G, st), "conv0")
    assert ld == 24 or float(Y[:, 24:].abs().max()) == 0.0
    want = nhwc(F.conv2d(x.double(), w0.double(), padding=1))
    close(Y[:, :24], want, what="conv0")
    s, q = fold_partials(part, G, 24)
    close(s, want.sum(0), what="conv0 sum", rtol=1e-6)
    close(q, (want * want).sum(0), what="conv0 sumsq", rtol=1e-6)


@pytest.mark.parametrize("B,H,W", [(2, 20, 44), (3, 17, 23), (2, 192, 256)])
def test_conv0_on_the_matrix_unit_is_bit_identical_to_the_valu_kernel(lib, B, H, W):
    """The MFMA adds the 27 terms of an output in order -- the VALU kernel's fma chain -- and the BatchNorm partial sums are
    formed in the association of that kernel's wave_sum: outputs AND partial sums bit for bit."""
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    x, w0 = torch.rand(B, 3, H, W, device=DEV), rnd(24, 3, 3, 3, scale=0.3)
    out = []
    for entry in ("eml_dense_conv0_fwd_f32", "eml_dense_conv0_fwd_mfma_f32"):
        Y = torch.zeros(B * H * W, 24, device=DEV)
        part = torch.zeros(G * 48, dtype=torch.float64, device=DEV)
        lib.check(getattr(L, entry)(p(x), p(w0), p(Y), 24, B, H, W, 24, p(part), G, st), entry)
        out.append((Y, part))
    assert torch.equal(out[0][0], out[1][0]), float((out[0][0] - out[1][0]).abs().max())
    assert torch.equal(out[0][1], out[1][1]), float((out[0][1] - out[1][1]).abs().max())


@pytest.mark.parametrize("relu,C", [(1, 24), (0, 171), (0, 150)])
def test_bn_apply_and_prepare(lib, relu, C):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    P, lds, ldd = 777, r16(C), r16(C) + 16
    src, dst = rnd(P, lds), torch.zeros(P, ldd, device=DEV)
    sc, sh = torch.rand(C, device=DEV) + 0.5, rnd(C, scale=0.3)
    part = torch.zeros(G * C * 2, dtype=torch.float64, device=DEV)
    lib.check(L.eml_dense_bn_apply_f32(p(src), lds, p(dst), ldd, C, P, p(sc), p(sh), relu, p(part), G, st), "apply")
    want = src[:, :C].double() * sc.double() + sh.double()
    if relu:
        want = want.clamp_min(0)
    close(dst[:, :C], want, what="bn_apply")
    assert float(dst[:, C:].abs().max()) == 0.0
    # bn_prepare folds those partials: mean / biased var / istd, scale / shift, running stats
    mean, var, istd = (torch.zeros(ldd, device=DEV) for _ in range(3))
    gamma, beta = torch.rand(C, device=DEV) + 0.5, rnd(C, scale=0.2)
    rm, rv = rnd(C, scale=0.1), torch.rand(C, device=DEV) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    Cpad = r16(C)
    scale, shift = torch.full((Cpad,), 9.0, device=DEV), torch.full((Cpad,), 9.0, device=DEV)
    lib.check(L.eml_dense_bn_prepare_f32(p(part), G, 2 * C, C, 0, float(P), p(mean), p(var), p(istd), p(gamma), p(beta),
                                         p(rm), p(rv), C, Cpad, 1e-5, 0.1, 1, p(scale), p(shift), st), "prepare")
    m, v = want.mean(0), want.var(0, unbiased=False)
    close(mean[:C], m, what="mean", atol=1e-6)
    close(var[:C], v, what="var", rtol=1e-5)
    close(istd[:C], 1 / torch.sqrt(v + 1e-5), what="istd", rtol=1e-5)
    close(scale[:C], gamma.double() / torch.sqrt(v + 1e-5), what="scale", rtol=1e-5)
    close(shift[:C], beta.double() - m * gamma.double() / torch.sqrt(v + 1e-5), what="shift", rtol=1e-5, atol=1e-5)
    assert float(scale[C:].abs().max() if Cpad > C else 0) == 0.0
    close(rm, 0.9 * rm0.double() + 0.1 * m, what="running_mean", atol=1e-6)
    close(rv, 0.9 * rv0.double() + 0.1 * want.var(0, unbiased=True), what="running_var", rtol=1e-5)
    # eval mode: affine from the running buffers, statistics untouched
    lib.check(L.eml_dense_bn_prepare_f32(None, 0, 0, 0, 0, float(P), p(mean), p(var), p(istd), p(gamma), p(beta), p(rm),
                                         p(rv), C, Cpad, 1e-5, 0.1, 0, p(scale), p(shift), st), "prepare eval")
    close(scale[:C], gamma.double() / torch.sqrt(rv.double() + 1e-5), what="eval scale", rtol=1e-5)


@pytest.mark.parametrize("pool,Cin,Cout,B,H,W", [(0, 24, 48, 2, 20, 44), (0, 330, 48, 1, 12, 20), (0, 150, 48, 3, 6, 10),
                                                 (1, 216, 108, 2, 12, 20), (1, 342, 171, 1, 8, 12),
                                                 (1, 300, 150, 2, 6, 10)])
def test_conv1x1_fwd(lib, pool, Cin, Cout, B, H, W):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    Kp, ld = r16(Cin), r16(Cin) + 16
    Pin = B * H * W
    P = Pin // 4 if pool else Pin
    X = rnd(Pin, ld)
    sc, sh = torch.zeros(Kp, device=DEV), torch.zeros(Kp, device=DEV)
    sc[:Cin], sh[:Cin] = torch.rand(Cin, device=DEV) + 0.5, rnd(Cin, scale=0.3)
    Wt = rnd(Cout, Cin, scale=1.0 / np.sqrt(Cin))
    nch = (Cout + 47) // 48
    Wp = torch.empty(nch * Kp * 48, device=DEV)
    lib.check(L.eml_dense_permute_w1_f32(p(Wt), Cout, Cin, Kp, p(Wp), st), "permute")
    ldo = r16(Cout)
    out = torch.zeros(P, ldo, device=DEV)
    part = torch.zeros(nch * G * 96, dtype=torch.float64, device=DEV)
    mask = None if pool else torch.zeros(((P + 255) // 256) * 16 * (Kp // 16) * 4, dtype=torch.int64, device=DEV)
    lib.check(L.eml_dense_conv1x1_fwd_f32(p(X), ld, P, H, W, pool, Kp, p(sc), p(sh), p(Wp), Cout, p(out), ldo, p(part),
                                          G, p(mask), st), "conv1x1")
    a = (X[:, :Cin].double() * sc[:Cin].double() + sh[:Cin].double()).clamp_min(0)
    if mask is not None:
        # relu_mask: word (pixel group pg, K-step j, t), bit r + 16*q <-> pixel 16*pg + r, channel 16*j + 4*q + t
        act = torch.zeros(((P + 255) // 256) * 256, Kp, dtype=torch.bool, device=DEV)
        act[:P, :Cin] = torch.addcmul(sh[:Cin], X[:, :Cin], sc[:Cin]) > 0     # f32 fma, like the kernel's bn_relu4
        bits = act.view(-1, 16, Kp // 16, 4, 4).permute(0, 2, 4, 3, 1).reshape(-1, 64)   # (pg, j, t, q*16 + r)
        got = ((mask.view(-1, 1) >> torch.arange(64, device=DEV)) & 1).bool()
        npg = (P + 15) // 16                                                # pixel groups past P hold clamped duplicates
        differ = (got.view(-1, (Kp // 16) * 4, 64)[:P // 16] != bits.view(-1, (Kp // 16) * 4, 64)[:P // 16])
        assert int(differ.sum()) <= 2, int(differ.sum())                    # (an fma within 1 ulp of 0 may round the other way)
        assert npg <= mask.numel() // ((Kp // 16) * 4)
    if pool:
        a = nhwc(F.avg_pool2d(nchw(a, B, H, W), 2, 2))
    want = a @ Wt.double().t()
    close(out[:, :Cout], want, what="conv1x1 out")
    assert float(out[:, Cout:].abs().max() if ldo > Cout else 0) == 0.0
    for ch in range(nch):
        nv = min(48, Cout - 48 * ch)
        s, q = fold_partials(part[ch * G * 96:], G, 48)
        close(s[:nv], want[:, 48 * ch:48 * ch + nv].sum(0), what="stats sum", rtol=1e-6, atol=1e-4)
        close(q[:nv], (want[:, 48 * ch:48 * ch + nv] ** 2).sum(0), what="stats sq", rtol=1e-5)


@pytest.mark.parametrize("B,H,W,c_out0,ld", [(2, 20, 44, 24, 64), (1, 8, 32, 150, 176), (3, 7, 9, 108, 128)])
def test_conv3x3_fwd(lib, B, H, W, c_out0, ld):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    P = B * H * W
    Z = rnd(P, 48)
    s2, t2 = torch.rand(48, device=DEV) + 0.5, rnd(48, scale=0.3)
    W2 = rnd(12, 48, 3, 3, scale=0.05)
    W2p = torch.empty(9 * 3 * 4 * 16 * 4, device=DEV)
    lib.check(L.eml_dense_permute_w2_f32(p(W2), 12, p(W2p), st), "permute2")
    X = torch.full((P, ld), 5.0, device=DEV)
    part = torch.zeros(G * 32, dtype=torch.float64, device=DEV)
    lib.check(L.eml_dense_conv3x3_fwd_f32(p(Z), p(s2), p(t2), p(W2p), p(X), ld, c_out0, B, H, W, p(part), G, st), "c3")
    zn = nchw(Z.double() * s2.double() + t2.double(), B, H, W)
    want = nhwc(F.conv2d(zn, W2.double(), padding=1))  # zero padding applies to the BN output
    close(X[:, c_out0:c_out0 + 12], want, what="conv3x3 out")
    assert bool((X[:, :c_out0] == 5.0).all()) and bool((X[:, c_out0 + 12:] == 5.0).all())
    s, q = fold_partials(part, G, 16)
    close(s[:12], want.sum(0), what="sum", rtol=1e-6, atol=1e-4)
    close(q[:12], (want ** 2).sum(0), what="sq", rtol=1e-5)


@pytest.mark.parametrize("B,H,W,c_out0,ld,band,grid", [
    (2, 20, 64, 24, 64, 7, 64),      # TX = 4, one wave: no edge exchange; ragged last band
    (1, 9, 80, 150, 176, 4, 64),     # TX = 5, one wave
    (2, 13, 160, 108, 128, 5, 3),    # TX = 5, two waves; fewer workgroups than bands (persistent loop)
    (1, 11, 320, 36, 224, 30, 64),   # TX = 5, four waves (block 1's width); one band taller than the image
    (2, 7, 256, 24, 64, 1, 64),      # TX = 4, four waves; one-row bands: every row is a top AND a bottom neighbour
    (3, 6, 128, 48, 64, 3, 7),       # TX = 4, two waves
])
def test_conv3x3_fwd_tap_packed(lib, B, H, W, c_out0, ld, band, grid):
    """eml_dense_conv3x3_fwd_tp_f32 (csrc/dense_fwd_tp.hip): the 9 taps x 12 channels as MFMA rows, the 3x3 sum as a
    shift-and-add of accumulators -- against F.conv2d in f64 and bit for bit against itself across band sizes."""
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    assert L.eml_dense_conv3x3_fwd_tp_supported(B, H, W) in (1, 2, 4)
    assert L.eml_dense_conv3x3_fwd_tp_supported(B, H, W + 8) == 0 and L.eml_dense_conv3x3_fwd_tp_supported(B, H, 48) == 0
    P = B * H * W
    Z = rnd(P, 48)
    s2, t2 = torch.rand(48, device=DEV) + 0.5, rnd(48, scale=0.3)
    W2 = rnd(12, 48, 3, 3, scale=0.05)
    W2t = torch.empty(7 * 3 * 64 * 4, device=DEV)
    lib.check(L.eml_dense_permute_w2_tp_f32(p(W2), p(W2t), st), "permute2 tp")
    outs = []
    for bd in (band, H):
        X = torch.full((P, ld), 5.0, device=DEV)
        part = torch.full((grid * 32,), 3.0, dtype=torch.float64, device=DEV)
        lib.check(L.eml_dense_conv3x3_fwd_tp_f32(p(Z), p(s2), p(t2), p(W2t), p(X), ld, c_out0, B, H, W, bd, p(part), grid,
                                                 st), "c3 tp")
        zn = nchw(Z.double() * s2.double() + t2.double(), B, H, W)
        want = nhwc(F.conv2d(zn, W2.double(), padding=1))  # zero padding applies to the BN output
        close(X[:, c_out0:c_out0 + 12], want, what="conv3x3 tp out")
        assert bool((X[:, :c_out0] == 5.0).all()) and bool((X[:, c_out0 + 12:] == 5.0).all())
        s, q = fold_partials(part, grid, 16)
        close(s[:12], want.sum(0), what="sum", rtol=1e-6, atol=1e-4)
        close(q[:12], (want ** 2).sum(0), what="sq", rtol=1e-5)
        assert float(s[12:].abs().max()) == 0.0 and float(q[12:].abs().max()) == 0.0
        outs.append(X[:, c_out0:c_out0 + 12].clone())
    assert torch.equal(outs[0], outs[1])   # the band split does not change a single bit
    # the batched re-layout (kind 3) writes the same W2t
    import ctypes
    dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("kind", "<i4"), ("Cout", "<i4"), ("Cin", "<i4"), ("Kp", "<i4"),
                   ("Ko", "<i4"), ("reserved", "<i4")])
    host = np.zeros(1, dtype=dt)
    dst = torch.full((7 * 3 * 64 * 4,), 7.0, device=DEV)
    host[0] = (W2.data_ptr(), dst.data_ptr(), 3, 12, 48, 48, 0, 0)
    descs = torch.from_numpy(host.view(np.uint8).copy()).to(DEV)
    lib.check(L.eml_dense_permute_batch_f32(p(descs), 1, st), "permute batch kind 3")
    assert torch.equal(dst, W2t)


def test_head_pool_fwd_bwd(lib):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    B, H, W, C, k, ld = 2, 8, 12, 171, 4, 176
    Fm = rnd(B * H * W, ld)
    out = torch.empty(B, C * (H // k) * (W // k), device=DEV)
    lib.check(L.eml_dense_head_pool_fwd_f32(p(Fm), ld, C, B, H, W, k, p(out), st), "head")
    f = nchw(Fm[:, :C].double(), B, H, W).requires_grad_(True)
    want = F.avg_pool2d(torch.relu(f), k).reshape(B, -1)
    close(out, want, what="head pool")
    gp = rnd(B, C * (H // k) * (W // k))
    dF = torch.zeros(B * H * W, ld, device=DEV)
    lib.check(L.eml_dense_head_pool_bwd_f32(p(gp), p(Fm), ld, C, B, H, W, k, p(dF), ld, st), "head bwd")
    (want * gp.double()).sum().backward()
    close(dF[:, :C], nhwc(f.grad), what="head pool bwd")


def test_permute_batch_equals_the_single_permutes(lib):
    """eml_dense_permute_batch_f32: every re-layout of a pass in one launch, bit for bit the three single kernels."""
    import ctypes
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("kind", "<i4"), ("Cout", "<i4"), ("Cin", "<i4"), ("Kp", "<i4"),
                   ("Ko", "<i4"), ("reserved", "<i4")])
    cases = [(0, 48, 36, 48, 0), (0, 171, 342, 352, 0), (1, 12, 48, 48, 0), (2, 48, 150, 160, 48), (2, 108, 216, 224, 112)]
    host = np.zeros(len(cases), dtype=dt)
    keep, want = [], []
    for i, (kind, cout, cin, kp, ko) in enumerate(cases):
        if kind == 1:
            Wt, n = rnd(12, 48, 3, 3), 9 * 3 * 4 * 16 * 4
        else:
            Wt, n = rnd(cout, cin), (((cout + 47) // 48) * kp * 48 if kind == 0 else kp * ko)
        dst, ref = torch.full((n,), 7.0, device=DEV), torch.empty(n, device=DEV)
        if kind == 0:
            lib.check(L.eml_dense_permute_w1_f32(p(Wt), cout, cin, kp, p(ref), st), "w1")
        elif kind == 1:
            lib.check(L.eml_dense_permute_w2_f32(p(Wt), 12, p(ref), st), "w2")
        else:
            lib.check(L.eml_dense_permute_w1_bwd_f32(p(Wt), cout, cin, kp, ko, p(ref), st), "w1 bwd")
        host[i] = (Wt.data_ptr(), dst.data_ptr(), kind, cout, cin, kp, ko, 0)
        keep.append((Wt, dst))
        want.append(ref)
    table = torch.from_numpy(host.view(np.uint8).copy()).to(DEV)
    lib.check(L.eml_dense_permute_batch_f32(p(table), len(cases), st), "batch")
    for (Wt, dst), ref in zip(keep, want):
        assert torch.equal(dst, ref)
    assert L.eml_dense_permute_batch_f32(None, 3, st) == -1 and L.eml_dense_permute_batch_f32(p(table), 0, st) == 0


@pytest.mark.parametrize("B,H,W,C,ld", [(2, 6, 10, 216, 224), (1, 4, 4, 20, 32)])
def test_pool_act(lib, B, H, W, C, ld):
    """A = mean2x2(relu(scale*x + shift)): the transition's operand (DenseNet.py:14-21, pool commuted before the conv)."""
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    Kp = (C + 15) // 16 * 16
    X = rnd(B * H * W, ld)
    sc, sh = torch.zeros(Kp, device=DEV), torch.zeros(Kp, device=DEV)
    sc[:C], sh[:C] = torch.rand(C, device=DEV) + 0.5, rnd(C, scale=0.3)
    A = torch.full((B * (H // 2) * (W // 2), Kp), 9.0, device=DEV)
    m16 = torch.zeros(B * (H // 2) * (W // 2) * (Kp // 4), dtype=torch.int16, device=DEV)
    lib.check(L.eml_dense_pool_act_f32(p(X), ld, B, H, W, Kp, p(sc), p(sh), p(A), Kp, p(m16), st), "pool_act")
    act = torch.relu(nchw(X[:, :Kp].double(), B, H, W) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    want = nhwc(F.avg_pool2d(act, 2))
    close(A, want, what="pooled activation", rtol=1e-6, atol=1e-6)
    assert float(A[:, C:].abs().max()) == 0.0 if Kp > C else True
    # relu_mask16: word [pooled pixel][quad q], bit 4*sub + g <-> window pixel sub (row-major), channel 4q + g
    on = (torch.addcmul(sh, X[:, :Kp], sc) > 0).view(B, H // 2, 2, W // 2, 2, Kp // 4, 4)      # f32 fma like the kernel
    bits = on.permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, 16)                                      # (b, oy, ox, q, dy, dx, g)
    got = ((m16.view(-1, 1).int() >> torch.arange(16, device=DEV)) & 1).bool()
    assert int((got != bits).sum()) <= 2
    A2 = torch.empty_like(A)
    lib.check(L.eml_dense_pool_act_f32(p(X), ld, B, H, W, Kp, p(sc), p(sh), p(A2), Kp, None, st), "pool_act (no mask)")
    assert torch.equal(A, A2)


# ------------------------------------------------------------------------------------------ backward
@pytest.mark.parametrize("B,H,W,c0,ld", [(2, 20, 44, 24, 64), (1, 8, 32, 162, 176), (3, 7, 9, 108, 128)])
def test_conv3x3_bwd_data_and_weight(lib, B, H, W, c0, ld):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    P = B * H * W
    Gd, Z = rnd(P, ld), rnd(P, 48)
    W2 = rnd(12, 48, 3, 3, scale=0.05)
    s2, t2 = torch.rand(48, device=DEV) + 0.5, rnd(48, scale=0.3)
    zmean, zistd = rnd(48, scale=0.1), torch.rand(48, device=DEV) + 0.5
    DZ = torch.empty(P, 48, device=DEV)
    part = torch.zeros(G * 96, dtype=torch.float64, device=DEV)
    # fused deferred affine: g = G + sB*X + sC on the layer's 12 channels, also written compactly to GF
    X = rnd(P, ld)
    sB, sC = rnd(ld, scale=0.3), rnd(ld, scale=0.3)
    GF = torch.full((P, 12), 7.0, device=DEV)
    lib.check(L.eml_dense_conv3x3_bwd_data_f32(p(Gd), ld, c0, p(W2), p(Z), p(zmean), p(zistd), p(DZ), B, H, W, p(part), G,
                                               p(X), ld, c0, p(sB), p(sC), p(GF), st), "c3 bwd data (fused affine)")
    gfull = Gd[:, c0:c0 + 12].double() + sB[c0:c0 + 12].double() * X[:, c0:c0 + 12].double() + sC[c0:c0 + 12].double()
    close(GF, gfull, what="GF", rtol=1e-6, atol=1e-6)
    zn0 = nchw(Z.double() * s2.double() + t2.double(), B, H, W).requires_grad_(True)
    (F.conv2d(zn0, W2.double(), padding=1) * nchw(gfull, B, H, W)).sum().backward()
    close(DZ, nhwc(zn0.grad), what="dzn (fused affine)")
    # the gradient slice taken from the compact (P,12) tensor of the narrow pass: g = N12 + sB*X + sC
    N12 = rnd(P, 12)
    lib.check(L.eml_dense_conv3x3_bwd_data_f32(p(N12), 12, 0, p(W2), p(Z), p(zmean), p(zistd), p(DZ), B, H, W, p(part), G,
                                               p(X), ld, c0, p(sB), p(sC), p(GF), st), "c3 bwd data (compact g)")
    gfull2 = N12.double() + sB[c0:c0 + 12].double() * X[:, c0:c0 + 12].double() + sC[c0:c0 + 12].double()
    close(GF, gfull2, what="GF (compact g)", rtol=1e-6, atol=1e-6)
    zn1 = nchw(Z.double() * s2.double() + t2.double(), B, H, W).requires_grad_(True)
    (F.conv2d(zn1, W2.double(), padding=1) * nchw(gfull2, B, H, W)).sum().backward()
    close(DZ, nhwc(zn1.grad), what="dzn (compact g)")
    # plain form (X == NULL)
    lib.check(L.eml_dense_conv3x3_bwd_data_f32(p(Gd), ld, c0, p(W2), p(Z), p(zmean), p(zistd), p(DZ), B, H, W, p(part), G,
                                               None, 0, 0, None, None, None, st), "c3 bwd data")
    zn = nchw(Z.double() * s2.double() + t2.double(), B, H, W).requires_grad_(True)
    w = W2.double().requires_grad_(True)
    g = nchw(Gd[:, c0:c0 + 12].double(), B, H, W)
    (F.conv2d(zn, w, padding=1) * g).sum().backward()
    dzn = nhwc(zn.grad)
    close(DZ, dzn, what="dzn")
    s1, s2p = fold_partials(part, G, 48)
    zh = (Z.double() - zmean.double()) * zistd.double()
    close(s1, dzn.sum(0), what="S1", rtol=1e-6, atol=1e-4)
    close(s2p, (dzn * zh).sum(0), what="S2", rtol=1e-6, atol=1e-4)
    partW = torch.empty(2 * G * 27 * 256, device=DEV)
    dW2 = torch.empty(12, 48, 3, 3, device=DEV)
    lib.check(L.eml_dense_conv3x3_bwd_weight_f32(p(Gd), ld, c0, p(Z), p(s2), p(t2), B, H, W, p(partW), p(dW2), G, st),
              "c3 bwd weight")
    close(dW2, w.grad, what="dW2", rtol=1e-4)
    Gc = Gd[:, c0:c0 + 12].contiguous()  # the compact (P,12) form the engine passes
    lib.check(L.eml_dense_conv3x3_bwd_weight_f32(p(Gc), 12, 0, p(Z), p(s2), p(t2), B, H, W, p(partW), p(dW2), G, st),
              "c3 bwd weight (compact g)")
    close(dW2, w.grad, what="dW2 (compact g)", rtol=1e-4)


@pytest.mark.parametrize("B,H,W,c0,ld,compact", [(2, 20, 44, 24, 64, False), (1, 8, 32, 108, 128, True), (3, 7, 9, 36, 64, True),
                                                   (4, 24, 64, 60, 224, False), (2, 17, 70, 120, 304, True),
                                                   (1, 8, 32, 162, 176, False), (2, 15, 20, 150, 352, True)])
def test_conv3x3_bwd_fused_equals_the_two_launches(lib, B, H, W, c0, ld, compact):
    """Round 4: data gradient (with the fused BN1 affine) + weight gradient of a layer in ONE pass over the tiles
    (conv3x3_bwd_fused_tp_kernel) against the two separate launches on the same buffers and the same grid: dzn and GF
    bitwise (same MFMA order, same tile order per workgroup), the BatchNorm statistics to f32 round-off of a 32-pixel row sum
    (they are reduced over the tile row before the f64 accumulation instead of after it), and all of it against f64 autograd.
    Ragged tiles (H, W not multiples of 8 / 32), the block gradient and the compact (P, 12) tensor as the g source."""
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    P = B * H * W
    Gd, Z, X = rnd(P, ld), rnd(P, 48), rnd(P, ld)
    N12 = rnd(P, 12)
    W2 = rnd(12, 48, 3, 3, scale=0.05)
    s2, t2 = torch.rand(48, device=DEV) + 0.5, rnd(48, scale=0.3)
    zmean, zistd = rnd(48, scale=0.1), torch.rand(48, device=DEV) + 0.5
    sB, sC = rnd(ld, scale=0.3), rnd(ld, scale=0.3)
    gsrc = (N12, 12, 0) if compact else (Gd, ld, c0)
    assert L.eml_dense_conv3x3_bwd_fused_supported(gsrc[1], gsrc[2], ld, c0) == 1
    out = {}
    for mode in ("two", "one"):
        DZ = torch.full((P, 48), 3.0, device=DEV)
        GF = torch.full((P, 12), 7.0, device=DEV)
        part = torch.zeros(G * 96, dtype=torch.float64, device=DEV)
        partW = torch.zeros(2 * G * 27 * 256, device=DEV)
        dW2 = torch.empty(12, 48, 3, 3, device=DEV)
        if mode == "two":
            lib.check(L.eml_dense_conv3x3_bwd_data_f32(p(gsrc[0]), gsrc[1], gsrc[2], p(W2), p(Z), p(zmean), p(zistd), p(DZ), B, H, W,
                                                       p(part), G, p(X), ld, c0, p(sB), p(sC), p(GF), st), "c3 bwd data")
            lib.check(L.eml_dense_conv3x3_bwd_weight_f32(p(GF), 12, 0, p(Z), p(s2), p(t2), B, H, W, p(partW), p(dW2), G, st),
                      "c3 bwd weight")
        else:
            lib.check(L.eml_dense_conv3x3_bwd_fused_f32(p(gsrc[0]), gsrc[1], gsrc[2], p(W2), p(Z), p(zmean), p(zistd), p(DZ), B, H,
                                                        W, p(part), G, p(X), ld, c0, p(sB), p(sC), p(GF), p(s2), p(t2), p(partW),
                                                        p(dW2), st), "c3 bwd fused")
        out[mode] = (DZ, GF, dW2, fold_partials(part, G, 48))
    for name, a, b in zip(("dzn", "GF"), out["one"][:2], out["two"][:2]):
        assert torch.equal(a, b), "%s differs from the separate launches: max %g" % (name, float((a - b).abs().max()))
    # round 6: the fused pass forms dW2 with the taps packed into the MFMA rows (21 accumulator tiles per wave, summed over the
    # eight waves, then over the workgroups): another summation order than the separate kernel's -- f32 round-off of a sum over P
    close(out["one"][2], out["two"][2], what="dW2 vs the separate launch", rtol=2e-5,
          atol=2e-5 * float(out["two"][2].abs().max()))
    for a, b, name in zip(out["one"][3], out["two"][3], ("S1", "S2")):
        close(a, b, what=name + " vs the separate launch", rtol=1e-6, atol=1e-5 * float(b.abs().max()))
    # and against f64 autograd
    g0 = gsrc[0][:, gsrc[2]:gsrc[2] + 12].double()
    gfull = g0 + sB[c0:c0 + 12].double() * X[:, c0:c0 + 12].double() + sC[c0:c0 + 12].double()
    zn = nchw(Z.double() * s2.double() + t2.double(), B, H, W).requires_grad_(True)
    w = W2.double().requires_grad_(True)
    (F.conv2d(zn, w, padding=1) * nchw(gfull, B, H, W)).sum().backward()
    dzn = nhwc(zn.grad)
    close(out["one"][0], dzn, what="dzn vs f64")
    close(out["one"][1], gfull, what="GF vs f64", rtol=1e-6, atol=1e-6)
    close(out["one"][2], w.grad, what="dW2 vs f64", rtol=1e-4)
    zh = (Z.double() - zmean.double()) * zistd.double()
    close(out["one"][3][0], dzn.sum(0), what="S1 vs f64", rtol=1e-6, atol=1e-4)
    close(out["one"][3][1], (dzn * zh).sum(0), what="S2 vs f64", rtol=1e-6, atol=1e-4)
    # 8-byte aligned channel offsets (block 3 of EMLight's encoder starts at channel 150) are taken with paired 8-byte loads;
    # odd ones are refused
    assert L.eml_dense_conv3x3_bwd_fused_supported(176, 162, 176, 162) == 1
    assert L.eml_dense_conv3x3_bwd_fused_supported(176, 161, 176, 161) == 0


def test_bn_bwd_finalize(lib):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    C, Cpad, R, n = 150, 160, 37, 1234.0
    part = torch.randn(R * 2 * C, dtype=torch.float64, device=DEV)
    gamma, mean, istd = torch.rand(C, device=DEV) + 0.5, rnd(C, scale=0.2), torch.rand(C, device=DEV) + 0.5
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    cA, cB, cC = (torch.full((Cpad,), 3.0, device=DEV) for _ in range(3))
    sB, sC = torch.ones(Cpad, device=DEV), torch.ones(Cpad, device=DEV)
    lib.check(L.eml_dense_bn_bwd_finalize_f32(p(part), R, 2 * C, n, p(gamma), p(mean), p(istd), C, Cpad, 1, p(dg), p(db),
                                              p(cA), p(cB), p(cC), p(sB), p(sC), 1, 0, Cpad, None, None, None, 0, None, None, st), "finalize")
    S = part.view(R, C, 2).sum(0)
    S1, S2 = S[:, 0], S[:, 1]
    ga, is_, mu = gamma.double(), istd.double(), mean.double()
    close(db, S1, what="dbeta")
    close(dg, S2, what="dgamma")
    close(cA[:C], ga * is_, what="cA")
    close(cB[:C], -ga * is_ * is_ * S2 / n, what="cB")
    close(cC[:C], -ga * is_ * S1 / n + ga * is_ * is_ * S2 / n * mu, what="cC", atol=1e-6)
    close(sB[:C], 1 + (-ga * is_ * is_ * S2 / n), what="sB accumulate")
    assert float(cA[C:].abs().max()) == 0.0 and float((sB[C:] - 1).abs().max()) == 0.0
    # channel-range form: only [40, 52) is touched
    cA2, sB2 = torch.full((Cpad,), 3.0, device=DEV), torch.ones(Cpad, device=DEV)
    dg2 = torch.full((C,), 7.0, device=DEV)
    lib.check(L.eml_dense_bn_bwd_finalize_f32(p(part), R, 2 * C, n, p(gamma), p(mean), p(istd), C, Cpad, 1, p(dg2), p(db),
                                              p(cA2), p(cB), p(cC), p(sB2), p(sC), 1, 40, 52, None, None, None, 0, None, None, st),
              "finalize range")
    close(dg2[40:52], S2[40:52], what="dgamma range")
    assert bool((dg2[:40] == 7.0).all()) and bool((dg2[52:] == 7.0).all())
    assert bool((cA2[:40] == 3.0).all()) and bool((sB2[52:] == 1.0).all())


def _dz(DY, Zr, cA, cB, cC, Cout):
    return cA[:Cout].double() * DY[:, :Cout].double() + cB[:Cout].double() * Zr[:, :Cout].double() + cC[:Cout].double()


@pytest.mark.parametrize("pool,Cin,Cout,B,H,W", [(0, 24, 48, 2, 20, 44), (0, 330, 48, 1, 12, 20), (0, 36, 48, 3, 6, 10),
                                                 (1, 216, 108, 2, 12, 20), (1, 342, 171, 1, 8, 12),
                                                 (1, 300, 150, 2, 6, 10)])
def test_conv1x1_bwd_weight_and_data(lib, pool, Cin, Cout, B, H, W):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    Kp, Ko = r16(Cin), r16(Cout)
    ld = Kp + 16
    Pin = B * H * W
    P = Pin // 4 if pool else Pin
    X = rnd(Pin, ld)
    s1, t1 = torch.zeros(Kp, device=DEV), torch.zeros(Kp, device=DEV)
    s1[:Cin], t1[:Cin] = torch.rand(Cin, device=DEV) + 0.5, rnd(Cin, scale=0.3)
    mean, istd = rnd(ld, scale=0.1), torch.rand(ld, device=DEV) + 0.5
    ld_dy = Ko + 16
    DY, Zr = rnd(P, ld_dy), rnd(P, Ko)
    cA, cB, cC = (torch.zeros(Ko, device=DEV) for _ in range(3))
    cA[:Cout], cB[:Cout], cC[:Cout] = rnd(Cout), rnd(Cout, scale=0.1), rnd(Cout, scale=0.1)
    Wt = rnd(Cout, Cin, scale=1.0 / np.sqrt(Cout))
    dz = _dz(DY, Zr, cA, cB, cC, Cout)
    pre = X[:, :Cin].double() * s1[:Cin].double() + t1[:Cin].double()
    a = pre.clamp_min(0)
    if pool:
        a = nhwc(F.avg_pool2d(nchw(a, B, H, W), 2, 2))
    # ---- weight gradient
    partW = torch.empty(G * Kp * 48, device=DEV)
    dW = torch.empty(Cout, Cin, device=DEV)
    lib.check(L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DY), ld_dy, p(Zr), Ko,
                                                 p(cA), p(cB), p(cC), Cout, p(partW), p(dW), G, None, None, 0, None, 0, None, None, st), "wgrad")
    close(dW, dz.t() @ a, what="dW", rtol=1e-4)
    if not pool and Cout == 48:   # dense layer: dz materialised for the data-gradient passes, separately and in place
        dz_out = torch.full((P, 48), 5.0, device=DEV)
        lib.check(L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DY), ld_dy, p(Zr), Ko,
                                                     p(cA), p(cB), p(cC), Cout, p(partW), p(dW), G, p(dz_out), None, 0, None, 0, None, None, st), "wgrad+dz")
        close(dW, dz.t() @ a, what="dW (dz_out)", rtol=1e-4)
        close(dz_out, dz, what="dz_out", rtol=1e-6, atol=1e-6)
        DYc = DY[:, :48].contiguous()
        lib.check(L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DYc), 48, p(Zr), Ko,
                                                     p(cA), p(cB), p(cC), Cout, p(partW), p(dW), G, p(DYc), None, 0, None, 0, None, None, st), "wgrad in place")
        close(dW, dz.t() @ a, what="dW (in place)", rtol=1e-4)
        close(DYc, dz, what="dz in place", rtol=1e-6, atol=1e-6)
        if Cin >= 24:
            # the narrow data pass riding on the weight-gradient kernel: N12 = G[:, k_lo:k_lo+12] + s1*mask*(dz W[:, k_lo:k_lo+12])
            k_lo = Cin - 12
            Gn = rnd(P, ld)
            N12 = torch.full((P, 12), 3.0, device=DEV)
            pn = torch.zeros(G * Kp * 2, dtype=torch.float64, device=DEV)
            dW_n, dz_n = torch.empty(Cout, Cin, device=DEV), torch.empty(P, 48, device=DEV)
            lib.check(L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DY), ld_dy, p(Zr), Ko,
                                                         p(cA), p(cB), p(cC), Cout, p(partW), p(dW_n), G, p(dz_n), p(Wt), k_lo,
                                                         p(Gn), ld, p(N12), p(pn), st), "wgrad + narrow")
            close(dW_n, dz.t() @ a, what="dW (with the narrow pass)", rtol=1e-4)
            close(dz_n, dz, what="dz_out (with the narrow pass)", rtol=1e-6, atol=1e-6)
            dam_n = torch.where(pre[:, k_lo:] > 0, dz @ Wt.double()[:, k_lo:], torch.zeros(P, 12, device=DEV, dtype=torch.float64))
            close(N12, Gn[:, k_lo:Cin].double() + s1[k_lo:Cin].double() * dam_n, what="fused N12", rtol=1e-4)
            S1n, S2n = fold_partials(pn, G, Kp)
            close(S1n[k_lo:Cin], dam_n.sum(0), what="fused narrow S1", rtol=1e-5, atol=1e-4)
            assert float(S2n.abs().max()) == 0.0 and float(S1n[:k_lo].abs().max()) == 0.0
            # the narrow pass's G operand as the COMPACT (P, 12) tensor of those channels (ldg == 12: the two-layer data
            # pass's `top` output): bit for bit the wide-row result
            Gc, N12c = Gn[:, k_lo:Cin].contiguous(), torch.full((P, 12), 5.0, device=DEV)
            pn2 = torch.zeros_like(pn)
            lib.check(L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DY), ld_dy, p(Zr), Ko,
                                                         p(cA), p(cB), p(cC), Cout, p(partW), p(dW_n), G, p(dz_n), p(Wt), k_lo,
                                                         p(Gc), 12, p(N12c), p(pn2), st), "wgrad + narrow, compact G")
            assert torch.equal(N12c, N12) and torch.equal(pn2, pn)
            # odd k_lo / range past Cin are refused
            assert L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DY), ld_dy, p(Zr), Ko,
                                                      p(cA), p(cB), p(cC), Cout, p(partW), p(dW_n), G, p(dz_n), p(Wt), k_lo + 1,
                                                      p(Gn), ld, p(N12), p(pn), st) == -1
    else:
        assert L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, H, W, pool, Kp, Cin, p(s1), p(t1), p(DY), ld_dy, p(Zr), Ko,
                                                  p(cA), p(cB), p(cC), Cout, p(partW), p(dW), G, p(DY), None, 0, None, 0, None, None, st) == -1
    # ---- data gradient, accumulate and overwrite modes, + BN1-backward partial sums
    Wd = torch.empty(Kp * Ko, device=DEV)
    lib.check(L.eml_dense_permute_w1_bwd_f32(p(Wt), Cout, Cin, Kp, Ko, p(Wd), st), "permute bwd")
    da = dz @ Wt.double()
    if pool:
        da = da.view(B, H // 2, 1, W // 2, 1, Cin).expand(B, H // 2, 2, W // 2, 2, Cin).reshape(Pin, Cin) * 0.25
    dam = torch.where(pre > 0, da, torch.zeros_like(da))
    xh = (X[:, :Cin].double() - mean[:Cin].double()) * istd[:Cin].double()
    for accumulate in (1, 0):
        G0 = rnd(Pin, ld)
        Gd = G0.clone()
        part = torch.zeros(G * Kp * 2, dtype=torch.float64, device=DEV)
        lib.check(L.eml_dense_conv1x1_bwd_data_f32(p(DY), ld_dy, p(Zr), Ko, p(cA), p(cB), p(cC), Ko, p(Wd), p(X), ld,
                                                   p(s1), p(t1), p(mean), p(istd), P, H, W, pool, Kp, p(Gd), ld,
                                                   accumulate, p(part), G, None, st), "dgrad")
        want = s1[:Cin].double() * dam + (G0[:, :Cin].double() if accumulate else 0)
        close(Gd[:, :Cin], want, what="G (accumulate=%d)" % accumulate, rtol=1e-4)
        assert torch.equal(Gd[:, Kp:], G0[:, Kp:])
        S1, S2 = fold_partials(part, G, Kp)
        close(S1[:Cin], dam.sum(0), what="S1", rtol=1e-5, atol=1e-4)
        close(S2[:Cin], (dam * xh).sum(0), what="S2", rtol=1e-5, atol=1e-4)
        if pool and Cout != 48:
            # the transition's x-free form: ReLU bits from pool_act, S1 only; BN's S2 from the conv's weight gradient
            A_ = torch.empty(P, Kp, device=DEV)
            m16 = torch.zeros(P * (Kp // 4), dtype=torch.int16, device=DEV)
            lib.check(L.eml_dense_pool_act_f32(p(X), ld, B, H, W, Kp, p(s1), p(t1), p(A_), Kp, p(m16), st), "pool_act mask")
            Gm = G0.clone()
            part.zero_()
            lib.check(L.eml_dense_conv1x1_bwd_data_f32(p(DY), ld_dy, p(Zr), Ko, p(cA), p(cB), p(cC), Ko, p(Wd), None, ld,
                                                       p(s1), p(t1), None, None, P, H, W, pool, Kp, p(Gm), ld,
                                                       accumulate, p(part), G, p(m16), st), "dgrad (masked)")
            close(Gm[:, :Cin], want, what="G masked (accumulate=%d)" % accumulate, rtol=1e-4)
            S1m, S2m = fold_partials(part, G, Kp)
            close(S1m[:Cin], dam.sum(0), what="S1 masked", rtol=1e-5, atol=1e-4)
            assert float(S2m.abs().max()) == 0.0
            # S2 = (sum_o W*dW - beta*S1) / gamma with scale1 = gamma*istd, shift1 = beta - mean*scale1
            gamma = (s1[:Cin] / istd[:Cin]).contiguous()
            beta = (t1[:Cin] + mean[:Cin] * s1[:Cin]).contiguous()
            dg, db = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
            lib.check(L.eml_dense_bn_bwd_finalize_f32(p(part), G, 2 * Kp, float(Pin), p(gamma), p(mean), p(istd), Cin, Kp, 1,
                                                      p(dg), p(db), None, None, None, None, None, 0, 0, Kp, p(beta), p(Wt),
                                                      p(dW), Cout, None, None, st), "finalize from dW")
            S2w = (dam * xh).sum(0)
            close(dg, S2w, what="dgamma from dW (transition)", rtol=5e-4, atol=5e-5 * max(float(S2w.abs().max()), 1.0))
    if not (pool and Cout != 48):
        assert L.eml_dense_conv1x1_bwd_data_f32(p(DY), ld_dy, p(Zr), Ko, p(cA), p(cB), p(cC), Ko, p(Wd), p(X), ld, p(s1), p(t1),
                                                p(mean), p(istd), P, H, W, pool, Kp, p(Gd), ld, 1, p(part), G, p(part), st) == -1


def test_grad_materialize_and_bn_bwd_stats(lib):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    P, ld, c0, n = 999, 64, 24, 12
    Gd, X = rnd(P, ld), rnd(P, ld)
    sB, sC = rnd(ld, scale=0.1), rnd(ld, scale=0.1)
    G0 = Gd.clone()
    lib.check(L.eml_dense_grad_materialize_f32(p(Gd), ld, p(X), ld, p(sB), p(sC), c0, n, P, st), "materialize")
    want = G0.double()
    want[:, c0:c0 + n] += sB[c0:c0 + n].double() * X[:, c0:c0 + n].double() + sC[c0:c0 + n].double()
    close(Gd, want, what="materialize")
    for relu, C in ((1, 24), (0, 171)):
        ldc = r16(C)
        DY, raw, out = rnd(P, ldc + 16), rnd(P, ldc), rnd(P, ldc)
        mean, istd = rnd(C, scale=0.1), torch.rand(C, device=DEV) + 0.5
        part = torch.zeros(G * C * 2, dtype=torch.float64, device=DEV)
        lib.check(L.eml_dense_bn_bwd_stats_f32(p(DY), ldc + 16, p(raw), ldc, p(out), ldc, relu, C, P, p(mean), p(istd),
                                               p(part), G, st), "bn bwd stats")
        dy = DY[:, :C].double()
        if relu:
            dy = torch.where(out[:, :C] > 0, dy, torch.zeros_like(dy))
        S1, S2 = fold_partials(part, G, C)
        close(S1, dy.sum(0), what="S1", atol=1e-4)
        close(S2, (dy * (raw[:, :C].double() - mean.double()) * istd.double()).sum(0), what="S2", atol=1e-4)


def test_conv0_bwd_weight(lib):
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    B, H, W, ld = 2, 20, 44, 32
    P = B * H * W
    x = torch.rand(B, 3, H, W, device=DEV)
    Gd, X1, Y0 = rnd(P, ld), rnd(P, ld), rnd(P, 24)
    cA, cB, cC = (torch.zeros(32, device=DEV) for _ in range(3))
    cA[:24], cB[:24], cC[:24] = rnd(24), rnd(24, scale=0.1), rnd(24, scale=0.1)
    partW = torch.empty(G * 4 * 1024, device=DEV)
    dW0 = torch.empty(24, 3, 3, 3, device=DEV)
    lib.check(L.eml_dense_conv0_bwd_weight_f32(p(x), p(Gd), ld, p(X1), ld, p(Y0), 24, p(cA), p(cB), p(cC), B, H, W,
                                               p(partW), p(dW0), G, st), "conv0 bwd")
    g = torch.where(X1[:, :24] > 0, Gd[:, :24].double(), torch.zeros(P, 24, device=DEV, dtype=torch.float64))
    dY0 = cA[:24].double() * g + cB[:24].double() * Y0.double() + cC[:24].double()
    w = torch.zeros(24, 3, 3, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, padding=1) * nchw(dY0, B, H, W)).sum().backward()
    close(dW0, w.grad, what="dW0", rtol=1e-4)


def test_norm0_backward_without_materialize_and_block_buffer(lib):
    """``eml_dense_norm0_bwd_stats_f32`` / ``eml_dense_conv0_bwd_weight_fused_f32`` (round 6) against the three launches they replace
    -- grad_materialize on block 1's input columns, then bn_bwd_stats / conv0_bwd_weight reading the block buffer: the
    statistics bit for bit (same expressions, same summation layout), dW0 bit for bit."""
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    B, H, W, ld, C = 3, 20, 44, 224, 24
    P = B * H * W
    x = torch.rand(B, 3, H, W, device=DEV)
    Y0 = rnd(P, C)
    s0, t0 = torch.rand(C, device=DEV) + 0.5, rnd(C, scale=0.3)
    X1 = torch.zeros(P, ld, device=DEV)
    fp = torch.zeros(G * C * 2, dtype=torch.float64, device=DEV)
    lib.check(L.eml_dense_bn_apply_f32(p(Y0), C, p(X1), ld, C, P, p(s0), p(t0), 1, p(fp), G, st), "bn_apply")   # the forward's x
    Gd = rnd(P, ld)
    sB, sC = rnd(ld, scale=0.1), rnd(ld, scale=0.1)
    mean, istd = rnd(C, scale=0.1), torch.rand(C, device=DEV) + 0.5
    cA, cB, cC = (torch.zeros(32, device=DEV) for _ in range(3))
    cA[:C], cB[:C], cC[:C] = rnd(C), rnd(C, scale=0.1), rnd(C, scale=0.1)
    partW = torch.empty(G * 4 * 1024, device=DEV)
    # the three launches
    Gm = Gd.clone()
    lib.check(L.eml_dense_grad_materialize_f32(p(Gm), ld, p(X1), ld, p(sB), p(sC), 0, C, P, st), "materialize")
    part_a = torch.zeros(G * C * 2, dtype=torch.float64, device=DEV)
    lib.check(L.eml_dense_bn_bwd_stats_f32(p(Gm), ld, p(Y0), C, p(X1), ld, 1, C, P, p(mean), p(istd), p(part_a), G, st), "stats")
    dW_a = torch.empty(C, 3, 3, 3, device=DEV)
    lib.check(L.eml_dense_conv0_bwd_weight_f32(p(x), p(Gm), ld, p(X1), ld, p(Y0), C, p(cA), p(cB), p(cC), B, H, W, p(partW), p(dW_a),
                                               G, st), "conv0 bwd")
    # the two fused ones, on the un-materialised G and without X1
    part_b = torch.zeros(G * C * 2, dtype=torch.float64, device=DEV)
    lib.check(L.eml_dense_norm0_bwd_stats_f32(p(Gd), ld, p(Y0), C, p(s0), p(t0), p(sB), p(sC), C, P, p(mean), p(istd), p(part_b), G,
                                              st), "norm0 stats")
    dW_b = torch.empty(C, 3, 3, 3, device=DEV)
    lib.check(L.eml_dense_conv0_bwd_weight_fused_f32(p(x), p(Gd), ld, p(Y0), C, p(s0), p(t0), p(sB), p(sC), p(cA), p(cB), p(cC), B, H,
                                                     W, p(partW), p(dW_b), G, st), "conv0 bwd fused")
    assert torch.equal(part_a, part_b) and torch.equal(dW_a, dW_b)
    # and against f64 torch
    xq = torch.clamp(Y0.double() * s0.double() + t0.double(), min=0)
    dy = torch.where(xq > 0, Gd[:, :C].double() + sB[:C].double() * xq + sC[:C].double(), torch.zeros_like(xq))
    S1, S2 = fold_partials(part_b, G, C)
    close(S1, dy.sum(0), what="S1", atol=1e-4)
    close(S2, (dy * (Y0.double() - mean.double()) * istd.double()).sum(0), what="S2", atol=1e-4)
    assert L.eml_dense_norm0_bwd_stats_f32(p(Gd), ld, p(Y0), C, p(s0), p(t0), p(sB), p(sC), 22, P, p(mean), p(istd), p(part_b), G,
                                           st) == -1                        # a channel count that is not a multiple of 4


@pytest.mark.parametrize("Cin_a,B,H,W", [(48, 2, 20, 44), (330, 1, 12, 20), (162, 3, 6, 10)])
def test_conv1x1_bwd_data_two_layers_per_pass(lib, Cin_a, B, H, W):
    """Layers (a, b = a-1) of a dense block: the narrow pass of layer a over b's 12 output channels, then
    the fused pass of both layers over [0, Cin_b) -- against the two single-layer results."""
    import ctypes
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    Cin_b = Cin_a - 12
    Kpa, Kpb = r16(Cin_a), r16(Cin_b)
    ld = Kpa + 16
    P = B * H * W
    X = rnd(P, ld)
    mean, istd = rnd(ld, scale=0.1), torch.rand(ld, device=DEV) + 0.5

    def layer(Cin, Kp):
        s1, t1 = torch.zeros(Kp, device=DEV), torch.zeros(Kp, device=DEV)
        s1[:Cin], t1[:Cin] = torch.rand(Cin, device=DEV) + 0.5, rnd(Cin, scale=0.3)
        d = dict(DZ=rnd(P, 48), Zr=rnd(P, 48), cA=rnd(48), cB=rnd(48, scale=0.1), cC=rnd(48, scale=0.1), s1=s1, t1=t1,
                 W=rnd(48, Cin, scale=0.15), Wd=torch.empty(Kp * 48, device=DEV),
                 part=torch.zeros(G * Kp * 2, dtype=torch.float64, device=DEV), Kp=Kp, Cin=Cin)
        lib.check(L.eml_dense_permute_w1_bwd_f32(p(d["W"]), 48, Cin, Kp, 48, p(d["Wd"]), st), "permute")
        dz = d["cA"].double() * d["DZ"].double() + d["cB"].double() * d["Zr"].double() + d["cC"].double()
        pre = X[:, :Cin].double() * s1[:Cin].double() + t1[:Cin].double()
        d["dzf"] = dz.float().contiguous()
        d["dam"] = torch.where(pre > 0, dz @ d["W"].double(), torch.zeros(P, Cin, device=DEV, dtype=torch.float64))
        return d
    A, Bl = layer(Cin_a, Kpa), layer(Cin_b, Kpb)
    xh = (X.double() - mean.double()) * istd.double()

    def run(layers, k_lo, k_hi, Gd, raw=False):
        arr = lambda key: (ctypes.c_void_p * len(layers))(*[y[key].data_ptr() for y in layers])
        if raw:   # DZ already holds the materialised dz: Zr == NULL
            args = (arr("dzf"), None, None, None, None)
        else:
            args = (arr("DZ"), arr("Zr"), arr("cA"), arr("cB"), arr("cC"))
        lib.check(L.eml_dense_conv1x1_bwd_data_multi_f32(
            len(layers), *args, arr("Wd"), arr("s1"), arr("t1"),
            arr("part"), (ctypes.c_int * len(layers))(*[y["Kp"] for y in layers]), p(X), ld, p(mean), p(istd), P, k_lo,
            k_hi, p(Gd), ld, G, None, st), "multi")

    G0 = rnd(P, ld)
    Gd = G0.clone()
    run([A], Cin_b, Cin_a, Gd)                    # narrow: only channels [Cin_b, Cin_a) of layer a
    want = G0.double()
    want[:, Cin_b:Cin_a] += A["s1"][Cin_b:Cin_a].double() * A["dam"][:, Cin_b:Cin_a]
    close(Gd, want, what="narrow G", rtol=1e-4)
    S1, S2 = fold_partials(A["part"], G, Kpa)
    close(S1[Cin_b:Cin_a], A["dam"][:, Cin_b:Cin_a].sum(0), what="narrow S1", rtol=1e-5, atol=1e-4)
    close(S2[Cin_b:Cin_a], (A["dam"] * xh[:, :Cin_a])[:, Cin_b:Cin_a].sum(0), what="narrow S2", rtol=1e-5, atol=1e-4)
    assert float(S1[:Cin_b].abs().max()) == 0.0
    run([A, Bl], 0, Cin_b, Gd)                    # fused: both layers over [0, Cin_b)
    want[:, :Cin_b] += A["s1"][:Cin_b].double() * A["dam"][:, :Cin_b] + Bl["s1"][:Cin_b].double() * Bl["dam"]
    close(Gd, want, what="fused G", rtol=1e-4)
    for y, Kp in ((A, Kpa), (Bl, Kpb)):
        S1, S2 = fold_partials(y["part"], G, Kp)
        close(S1[:Cin_b], y["dam"][:, :Cin_b].sum(0), what="fused S1", rtol=1e-5, atol=1e-4)
        close(S2[:Cin_b], (y["dam"][:, :Cin_b] * xh[:, :Cin_b]).sum(0), what="fused S2", rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("Cin_a,B,H,W", [(48, 2, 20, 44), (330, 1, 12, 20), (162, 3, 6, 10), (174, 2, 9, 7)])
def test_conv1x1_bwd_narrow_and_raw_dz_passes(lib, Cin_a, B, H, W):
    """The engine's pair schedule on materialised dz: the narrow pass of layer a as a compact (P,12) increment
    (eml_dense_conv1x1_bwd_narrow_f32; channel offsets with k_lo % 4 == 2 occur in block 3), then the fused pass of both
    layers with Zr == NULL -- against f64 torch."""
    import ctypes
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    Cin_b = Cin_a - 12
    Kpa, Kpb = r16(Cin_a), r16(Cin_b)
    ld = Kpa + 16
    P = B * H * W
    X = rnd(P, ld)
    mean, istd = rnd(ld, scale=0.1), torch.rand(ld, device=DEV) + 0.5
    xh = (X.double() - mean.double()) * istd.double()

    def layer(Cin, Kp):
        s1, t1 = torch.zeros(Kp, device=DEV), torch.zeros(Kp, device=DEV)
        s1[:Cin], t1[:Cin] = torch.rand(Cin, device=DEV) + 0.5, rnd(Cin, scale=0.3)
        d = dict(dz=rnd(P, 48), s1=s1, t1=t1, W=rnd(48, Cin, scale=0.15), Wd=torch.empty(Kp * 48, device=DEV),
                 part=torch.zeros(G * Kp * 2, dtype=torch.float64, device=DEV), Kp=Kp, Cin=Cin)
        lib.check(L.eml_dense_permute_w1_bwd_f32(p(d["W"]), 48, Cin, Kp, 48, p(d["Wd"]), st), "permute")
        pre = X[:, :Cin].double() * s1[:Cin].double() + t1[:Cin].double()
        d["dam"] = torch.where(pre > 0, d["dz"].double() @ d["W"].double(), torch.zeros(P, Cin, device=DEV, dtype=torch.float64))
        return d
    A, Bl = layer(Cin_a, Kpa), layer(Cin_b, Kpb)
    N12 = torch.full((P, 12), 3.0, device=DEV)
    G0 = rnd(P, ld)
    Gd = G0.clone()
    lib.check(L.eml_dense_conv1x1_bwd_narrow_f32(p(A["dz"]), p(A["W"]), Cin_a, Cin_b, p(X), ld, p(A["s1"]), p(A["t1"]), p(mean),
                                                 p(istd), P, p(Gd), ld, p(N12), p(A["part"]), Kpa, G, st), "narrow")
    close(N12, G0[:, Cin_b:Cin_a].double() + A["s1"][Cin_b:Cin_a].double() * A["dam"][:, Cin_b:Cin_a], what="N12", rtol=1e-4)
    assert torch.equal(Gd, G0)   # the block gradient itself is only read
    S1, S2 = fold_partials(A["part"], G, Kpa)
    close(S1[Cin_b:Cin_a], A["dam"][:, Cin_b:Cin_a].sum(0), what="narrow S1", rtol=1e-5, atol=1e-4)
    close(S2[Cin_b:Cin_a], (A["dam"] * xh[:, :Cin_a])[:, Cin_b:Cin_a].sum(0), what="narrow S2", rtol=1e-5, atol=1e-4)
    assert float(S1[:Cin_b].abs().max()) == 0.0   # nothing else of the partial rows is written
    layers = [A, Bl]
    arr = lambda key: (ctypes.c_void_p * 2)(*[y[key].data_ptr() for y in layers])
    lib.check(L.eml_dense_conv1x1_bwd_data_multi_f32(
        2, arr("dz"), None, None, None, None, arr("Wd"), arr("s1"), arr("t1"), arr("part"),
        (ctypes.c_int * 2)(Kpa, Kpb), p(X), ld, p(mean), p(istd), P, 0, Cin_b, p(Gd), ld, G, None, st), "multi raw")
    want = G0.double()
    want[:, :Cin_b] += A["s1"][:Cin_b].double() * A["dam"][:, :Cin_b] + Bl["s1"][:Cin_b].double() * Bl["dam"]
    close(Gd, want, what="fused G (raw dz)", rtol=1e-4)
    for y, Kp in ((A, Kpa), (Bl, Kpb)):
        S1, S2 = fold_partials(y["part"], G, Kp)
        close(S1[:Cin_b], y["dam"][:, :Cin_b].sum(0), what="fused S1", rtol=1e-5, atol=1e-4)
        close(S2[:Cin_b], (y["dam"][:, :Cin_b] * xh[:, :Cin_b]).sum(0), what="fused S2", rtol=1e-5, atol=1e-4)
    # bad arguments: odd channel offset, range past Cin
    assert L.eml_dense_conv1x1_bwd_narrow_f32(p(A["dz"]), p(A["W"]), Cin_a, Cin_b + 1, p(X), ld, p(A["s1"]), p(A["t1"]), p(mean),
                                              p(istd), P, p(Gd), ld, p(N12), p(A["part"]), Kpa, G, st) == -1
    assert L.eml_dense_conv1x1_bwd_narrow_f32(p(A["dz"]), p(A["W"]), Cin_a, Cin_a - 2, p(X), ld, p(A["s1"]), p(A["t1"]), p(mean),
                                              p(istd), P, p(Gd), ld, p(N12), p(A["part"]), Kpa, G, st) == -1


@pytest.mark.parametrize("Cin_a,B,H,W", [(48, 2, 16, 32), (330, 1, 16, 16), (162, 3, 16, 16), (174, 2, 9, 7),
                                         (216, 2, 16, 16), (132, 3, 9, 7)])
def test_conv1x1_bwd_masked_pass_and_bn1_from_weight_gradient(lib, Cin_a, B, H, W):
    """The X-free data-gradient pass: the forward kernel's ReLU bits replace relu(bn1(x)) > 0, the pass accumulates S1
    only, and BN1's S2 = sum dy*xhat is recovered from the conv's weight gradient:
    S2 = (sum_o W[o][c]*dW[o][c] - beta*S1) / gamma -- all against f64 torch on the same inputs."""
    import ctypes
    L, p, st = lib.lib(), lib.ptr, lib.current_stream()
    Cin_b = Cin_a - 12
    Kpa, Kpb = r16(Cin_a), r16(Cin_b)
    ld = Kpa + 16
    P = B * H * W
    X = rnd(P, ld)
    mu_t = X[:, :Cin_a].double().mean(0)
    var_t = X[:, :Cin_a].double().var(0, unbiased=False)
    mean, istd = torch.zeros(ld, device=DEV), torch.ones(ld, device=DEV)
    mean[:Cin_a], istd[:Cin_a] = mu_t.float(), (var_t + 1e-5).rsqrt().float()
    xh = (X.double() - mean.double()) * istd.double()

    def layer(Cin, Kp):
        gamma, beta = torch.rand(Cin, device=DEV) + 0.5, rnd(Cin, scale=0.3)
        s1, t1 = torch.zeros(Kp, device=DEV), torch.zeros(Kp, device=DEV)
        s1[:Cin] = gamma * istd[:Cin]
        t1[:Cin] = beta - mean[:Cin] * s1[:Cin]
        d = dict(dz=rnd(P, 48), s1=s1, t1=t1, gamma=gamma, beta=beta, W=rnd(48, Cin, scale=0.15),
                 Wd=torch.empty(Kp * 48, device=DEV), Wp=torch.empty(Kp * 48, device=DEV),
                 part=torch.zeros(G * Kp * 2, dtype=torch.float64, device=DEV), Kp=Kp, Cin=Cin,
                 mask=torch.zeros(((P + 255) // 256) * 16 * (Kp // 16) * 4, dtype=torch.int64, device=DEV))
        lib.check(L.eml_dense_permute_w1_bwd_f32(p(d["W"]), 48, Cin, Kp, 48, p(d["Wd"]), st), "permute bwd")
        lib.check(L.eml_dense_permute_w1_f32(p(d["W"]), 48, Cin, Kp, p(d["Wp"]), st), "permute fwd")
        z, fp = torch.empty(P, 48, device=DEV), torch.zeros(G * 96, dtype=torch.float64, device=DEV)
        lib.check(L.eml_dense_conv1x1_fwd_f32(p(X), ld, P, H, W, 0, Kp, p(s1), p(t1), p(d["Wp"]), 48, p(z), 48, p(fp), G,
                                              p(d["mask"]), st), "forward (writes the mask)")
        pre = torch.addcmul(t1[:Cin], X[:, :Cin], s1[:Cin])              # f32, the forward's own expression
        a = pre.double().clamp_min(0)
        d["dam"] = torch.where(pre > 0, d["dz"].double() @ d["W"].double(), torch.zeros(P, Cin, device=DEV, dtype=torch.float64))
        d["dW"] = (d["dz"].double().t() @ a).float().contiguous()      # the conv's weight gradient (48, Cin)
        return d
    A, Bl = layer(Cin_a, Kpa), layer(Cin_b, Kpb)
    layers = [A, Bl]
    arr = lambda key: (ctypes.c_void_p * 2)(*[y[key].data_ptr() for y in layers])
    G0 = rnd(P, ld)
    Gd = G0.clone()
    lib.check(L.eml_dense_conv1x1_bwd_data_multi_f32(
        2, arr("dz"), None, None, None, None, arr("Wd"), arr("s1"), arr("t1"), arr("part"),
        (ctypes.c_int * 2)(Kpa, Kpb), None, ld, None, None, P, 0, Cin_b, p(Gd), ld, G, arr("mask"), st), "multi masked")
    want = G0.double()
    want[:, :Cin_b] += A["s1"][:Cin_b].double() * A["dam"][:, :Cin_b] + Bl["s1"][:Cin_b].double() * Bl["dam"]
    close(Gd, want, what="fused G (masked)", rtol=1e-4)
    n = float(P)
    for y, Kp in ((A, Kpa), (Bl, Kpb)):
        S1, S2z = fold_partials(y["part"], G, Kp)
        close(S1[:Cin_b], y["dam"][:, :Cin_b].sum(0), what="masked S1", rtol=1e-5, atol=1e-4)
        assert float(S2z.abs().max()) == 0.0                            # the S2 slots stay zero
        C = y["Cin"]
        dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        lib.check(L.eml_dense_bn_bwd_finalize_f32(p(y["part"]), G, 2 * Kp, n, p(y["gamma"]), p(mean), p(istd), C, Kp, 1, p(dg),
                                                  p(db), None, None, None, None, None, 0, 0, Cin_b, p(y["beta"]), p(y["W"]),
                                                  p(y["dW"]), 48, None, None, st), "finalize from dW")
        S2 = (y["dam"][:, :Cin_b] * xh[:, :Cin_b]).sum(0)
        scale = float(S2.abs().max())
        close(dg[:Cin_b], S2, what="dgamma from the weight gradient", rtol=2e-4, atol=2e-5 * max(scale, 1.0))
        close(db[:Cin_b], y["dam"][:, :Cin_b].sum(0), what="dbeta", rtol=1e-5, atol=1e-4)
    # the same pass with its top 24 columns leaving as two compact (P, 12) tensors: bit for bit the values the plain pass
    # writes into G, G's own top 24 columns untouched, the statistics identical
    if Cin_b % 4 == 0:
        Gt, top = G0.clone(), torch.full((2, P, 12), float("nan"), device=DEV)
        parts = [y["part"].clone() for y in layers]
        for y in layers:
            y["part"].zero_()
        lib.check(L.eml_dense_conv1x1_bwd_data_multi_top_f32(
            arr("dz"), arr("Wd"), arr("s1"), arr("t1"), arr("part"), (ctypes.c_int * 2)(Kpa, Kpb), P, Cin_b, p(Gt), ld, G,
            arr("mask"), p(top), st), "multi masked, compact top")
        assert torch.equal(Gt[:, :Cin_b - 24], Gd[:, :Cin_b - 24]) and torch.equal(Gt[:, Cin_b - 24:], G0[:, Cin_b - 24:])
        assert torch.equal(top[0], Gd[:, Cin_b - 24:Cin_b - 12]) and torch.equal(top[1], Gd[:, Cin_b - 12:Cin_b])
        for y, q in zip(layers, parts):
            assert torch.equal(y["part"], q)
        assert L.eml_dense_conv1x1_bwd_data_multi_top_f32(
            arr("dz"), arr("Wd"), arr("s1"), arr("t1"), arr("part"), (ctypes.c_int * 2)(Kpa, Kpb), P, Cin_b - 2, p(Gt), ld, G,
            arr("mask"), p(top), st) == -1                                # a range that would split a channel quad
    # single-layer form
    Gd1 = G0.clone()
    A["part"].zero_()
    one = lambda key: (ctypes.c_void_p * 1)(A[key].data_ptr())
    lib.check(L.eml_dense_conv1x1_bwd_data_multi_f32(
        1, one("dz"), None, None, None, None, one("Wd"), one("s1"), one("t1"), one("part"), (ctypes.c_int * 1)(Kpa), None, ld,
        None, None, P, 0, Cin_a, p(Gd1), ld, G, one("mask"), st), "single masked")
    want1 = G0.double()
    want1[:, :Cin_a] += A["s1"][:Cin_a].double() * A["dam"]
    close(Gd1, want1, what="single-layer G (masked)", rtol=1e-4)
    # masks without the materialised dz are refused
    assert L.eml_dense_conv1x1_bwd_data_multi_f32(
        1, one("dz"), one("dz"), one("s1"), one("s1"), one("s1"), one("Wd"), one("s1"), one("t1"), one("part"),
        (ctypes.c_int * 1)(Kpa), None, ld, None, None, P, 0, Cin_a, p(Gd1), ld, G, one("mask"), st) == -1
