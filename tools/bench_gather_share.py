"""Per-layer A/B of the row-shared-corner gather (EML_TAP_ROWSHARE, csrc/gather_gemm2.h, round 5): the generator's fused
SphereConv shapes at B = 32, forward / SPADE epilogue / input gradient, 16-load kernel (flags 0) against the 10-load kernel
(flags 1) in ONE process on one box; outputs compared bit for bit.  Also times the spectral-norm launches of the generator's
weight shapes (csrc/spectral.hip).   python tools/bench_gather_share.py [B] > profiles/r05_gather_share.jsonl"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib
from emlight_amd.GenProjector import spherenet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PROBE = len(sys.argv) > 2 and sys.argv[2] == "probe"   # counters run: one layer, 16-load / 10-load / all-taps-to-pixel-0 only
dev = torch.device("cuda")
L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()


def events(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


LAYERS = [("up_3 gamma|beta 128->256 @128x256", 128, 256, 128, 256, "spade"), ("up_3 gamma|beta 128->128 @128x256", 128, 128, 128, 256, "spade"),
          ("up_3 conv_0 128->64 @128x256", 128, 64, 128, 256, "fwd"), ("up_3 conv_1 64->64 @128x256", 64, 64, 128, 256, "fwd"),
          ("up_2 gamma|beta 128->512 @64x128", 128, 512, 64, 128, "spade"), ("up_2 conv_0 256->128 @64x128", 256, 128, 64, 128, "fwd"),
          ("up_2 conv_1 128->128 @64x128", 128, 128, 64, 128, "fwd"), ("up_1 conv_0 512->256 @32x64", 512, 256, 32, 64, "fwd"),
          ("up_3 conv_0 dgrad 128<-64 @128x256", 128, 64, 128, 256, "dgrad"), ("up_2 conv_0 dgrad 256<-128 @64x128", 256, 128, 64, 128, "dgrad"),
          ("up_3 gamma|beta dgrad 128<-256 @128x256", 128, 256, 128, 256, "dgrad")]
if PROBE:
    LAYERS = [("up_3 gamma|beta 128->128 @128x256", 128, 128, 128, 256, "fwd")]
for name, C, O, H, W, role in LAYERS:
    geo = spherenet.sphere_geometry(H, W, 1, dev)
    po = H * W
    M = B * po
    gflop = 2.0 * M * 9 * C * O / 1e9
    row = {"layer": name, "role": role, "B": B, "gflop": round(gflop, 1), "rowshare": geo.rowshare}
    if role == "dgrad":
        tidx, twgt, rowmax, ke = geo.transposed_table()
        row.update(ke=ke, t_rowshare=geo.t_rowshare)
        gy = torch.randn(M, O, device=dev)
        w2t = torch.randn(C, 9 * O, device=dev) * 0.05
        outs = {}
        for flags in (0, 1) if (ke == 4 and geo.t_rowshare) else (0,):
            gx = torch.empty(M, C, device=dev)
            fn = lambda: _lib.check(L.eml_sphere_conv_dgrad_fused_f32(p(gy), p(tidx), p(twgt), p(rowmax), ke, p(w2t), p(gx), B, po, po,
                                                                      C, O, flags, st), "dgrad")
            t = events(fn)
            row["flags%d" % flags] = {"ms": round(t, 4), "tflops": round(gflop / t, 1)}
            outs[flags] = gx
        if len(outs) == 2:
            row["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
    else:
        x = torch.randn(M, C, device=dev)
        w2 = torch.randn(O, 9 * C, device=dev) * 0.05
        bias = torch.randn(O, device=dev)
        outs = {}
        for flags in (0, 1):
            if role == "spade":
                Cn = O // 2
                g_ = torch.Generator(device=dev).manual_seed(5)   # the same operands for both kernels
                xn = torch.randn(M, Cn, device=dev, generator=g_)
                mean, istd = torch.randn(Cn, device=dev, generator=g_), torch.rand(Cn, device=dev, generator=g_) + 0.5
                y = torch.empty(M, Cn, device=dev)
                g = torch.empty(M, Cn, device=dev)
                fn = lambda: _lib.check(L.eml_sphere_conv_spade_fwd_f32(p(x), p(geo.idx), p(geo.wgt), p(w2), p(bias), p(xn), p(mean),
                                                                        p(istd), p(y), p(g), B, H, W, C, Cn, 0, 0.2, flags, st), "spade")
            else:
                y = torch.empty(M, O, device=dev)
                fn = lambda: _lib.check(L.eml_sphere_conv_fwd_fused_ex_f32(p(x), p(geo.idx), p(geo.wgt), p(w2), p(bias), p(y), B, po, po,
                                                                           C, O, 4, None, 0.2, flags, st), "fwd")
            t = events(fn)
            row["flags%d" % flags] = {"ms": round(t, 4), "tflops": round(gflop / t, 1)}
            outs[flags] = y
        row["bit_identical"] = bool(torch.equal(outs[0], outs[1]))
    print(json.dumps(row), flush=True)

# what do the gathered loads cost when they cannot miss?  The same kernels on a table whose every entry is pixel 0 of the sample
# (one 512-byte line per sample: L1 hits): if the rate jumps, the gather is bound by memory latency / L2 traffic, not by issue
for name, C, O, H, W in (("128->128 @128x256, all taps -> pixel 0", 128, 128, 128, 256), ("256->128 @64x128, all taps -> pixel 0", 256, 128, 64, 128))[:1 if PROBE else 2]:
    geo = spherenet.sphere_geometry(H, W, 1, dev)
    po = H * W
    M = B * po
    gflop = 2.0 * M * 9 * C * O / 1e9
    idx0 = torch.zeros_like(geo.idx)
    x = torch.randn(M, C, device=dev)
    w2 = torch.randn(O, 9 * C, device=dev) * 0.05
    y = torch.empty(M, O, device=dev)
    row = {"layer": name, "role": "fwd-local", "B": B, "gflop": round(gflop, 1)}
    for flags in (0, 1):
        fn = lambda: _lib.check(L.eml_sphere_conv_fwd_fused_ex_f32(p(x), p(idx0), p(geo.wgt), p(w2), None, p(y), B, po, po, C, O, 4,
                                                                   None, 1.0, flags, st), "fwd")
        t = events(fn)
        row["flags%d" % flags] = {"ms": round(t, 4), "tflops": round(gflop / t, 1)}
    print(json.dumps(row), flush=True)

# spectral norm: the forward's five launches for the generator's weight shapes (training mode: one power iteration)
for O, C in () if PROBE else ((1024, 1024), (512, 1024), (512, 512), (256, 512), (256, 256), (128, 256), (128, 128), (64, 128), (64, 64)):
    w = torch.randn(O, C, 3, 3, device=dev) * 0.02
    u, v = torch.randn(O, device=dev), torch.randn(9 * C, device=dev)
    w2 = torch.empty(O, 9 * C, device=dev)
    sigma = torch.empty(1, device=dev)
    uv = torch.empty(O + 9 * C, device=dev)
    scratch = torch.empty(L.eml_spectral_norm_scratch_floats(O, C), device=dev)
    fn = lambda: _lib.check(L.eml_spectral_norm_w2_f32(p(w), p(u), p(v), 1, 1e-12, p(w2), p(sigma), p(uv), p(scratch), O, C, st), "sn")
    t = events(fn, reps=50)
    mb = O * 9 * C * 4 / 1e6
    print(json.dumps({"spectral_norm_w2": [O, C], "weight_MB": round(mb, 1), "us": round(t * 1e3, 1),
                      "GB_s_on_4_passes_of_W": round(4 * mb / t, 1)}), flush=True)
