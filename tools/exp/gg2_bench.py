"""A/B of the gather-GEMM kernels on the projector's layer shapes (B = 32): round 2's kernel (libemlight_hip.so) against the
configurations of gather_gemm2_kernel in build_exp/libgg2_exp*.so (tools/exp/gg2_exp.hip).  Forward role on the sphere
table, the input-gradient role on the transposed table (ke = 4 / 8, pole rows), the planar (ke = 1) role of the VGG stack.
    python tools/exp/gg2_bench.py [reps]      -> one JSON line per (shape, role) with TF/s per variant and max rel. error"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from emlight_amd import _lib  # noqa: E402
from emlight_amd.GenProjector.spherenet import sphere_geometry  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
LIBS = {}
import glob
ONLY = os.environ.get("GG2_ONLY")          # substring of the layer names to run (PMC passes time one shape)
ZERO = os.environ.get("GG2_ZERO") == "1"   # zero-filled operands: the clock the part reaches without data toggling
for path in sorted(glob.glob(os.path.join(ROOT, "build_exp", "libgg2_exp*.so"))):
    name = os.path.basename(path)
    if True:
        h = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        h.gg2_exp_fwd.restype = ctypes.c_int
        h.gg2_exp_fwd.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, vp] + [ctypes.c_int] * 6 + [vp, vp, ctypes.c_float, vp]
        LIBS[name.replace("libgg2_exp", "v2").replace(".so", "")] = h
L, p = _lib.lib(), _lib.ptr


def events(fn, reps=REPS):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, B, C, O, H, W, stride=1, role="fwd", kind="sphere"):
    if ONLY and ONLY not in name + " " + role:
        return
    dev = torch.device("cuda")
    geo = sphere_geometry(H, W, stride, dev, kind)
    po, hw = geo.ho * geo.wo, H * W
    st = _lib.current_stream()
    torch.manual_seed(1)
    bias = torch.randn(O if role != "dgrad" else C, device=dev)
    if role == "dgrad":      # rows gathered: dY (B*po, O); destination: the HW input pixels; "C" of the kernel = O, "O" = C
        tidx, twgt, rowmax, ke = geo.transposed_table()
        src = torch.randn(B * po, O, device=dev)
        w2 = torch.randn(C, 9 * O, device=dev) * 0.05
        args = dict(X=src, idx=tidx, wgt=twgt, W2=w2, B=B, HW=po, Po=hw, C=O, O=C, ke=ke, rowmax=rowmax)
    else:
        tab = (geo.idx1, geo.wgt1, 1) if (kind == "planar") else (geo.idx, geo.wgt, 4)
        src = torch.randn(B * hw, C, device=dev)
        w2 = torch.randn(O, 9 * C, device=dev) * 0.05
        args = dict(X=src, idx=tab[0], wgt=tab[1], W2=w2, B=B, HW=hw, Po=po, C=C, O=O, ke=tab[2], rowmax=None)
    a = args
    if ZERO:
        a["X"].zero_()
        a["W2"].zero_()
    M = a["B"] * a["Po"]
    y_old = torch.empty(M, a["O"], device=dev)
    gflop = 2.0 * M * 9 * a["C"] * a["O"] / 1e9

    def old():
        if role == "dgrad":
            _lib.check(L.eml_sphere_conv_dgrad_fused_f32(p(a["X"]), p(a["idx"]), p(a["wgt"]), p(a["rowmax"]), a["ke"], p(a["W2"]),
                                                         p(y_old), B, hw, po, C, O, st), "old dgrad")
        else:
            _lib.check(L.eml_sphere_conv_fwd_fused_ex_f32(p(a["X"]), p(a["idx"]), p(a["wgt"]), p(a["W2"]), p(bias), p(y_old), B, hw,
                                                          po, C, O, a["ke"], None, 0.2, st), "old fwd")
    row = {"layer": name, "role": role, "ke": a["ke"], "M": M, "C": a["C"], "O": a["O"], "gflop": round(gflop, 1)}
    t = events(old)
    row["old"] = {"ms": round(t, 4), "tflops": round(gflop / t, 1)}
    scale = float(y_old.abs().max()) + 1e-30
    for lname, h in LIBS.items():
        for var in (0, 2, 3, 5, 6):
            if var == 3 and a["O"] % 256:
                continue
            if lname != "v2" and var not in (0, 5):
                continue
            y = torch.zeros(M, a["O"], device=dev)

            def new():
                rc = h.gg2_exp_fwd(var, p(a["X"]), p(a["idx"]), p(a["wgt"]), p(a["W2"]), p(bias) if role != "dgrad" else None, p(y),
                                   a["B"], a["HW"], a["Po"], a["C"], a["O"], a["ke"], p(a["rowmax"]), None,
                                   ctypes.c_float(0.2 if role != "dgrad" else 1.0), st)
                assert rc == 0, rc
            try:
                t = events(new)
                err = float((y - y_old).abs().max()) / scale
                row["%s.%d" % (lname, var)] = {"ms": round(t, 4), "tflops": round(gflop / t, 1), "relerr": float("%.2e" % err)}
            except Exception as e:   # noqa: BLE001
                row["%s.%d" % (lname, var)] = {"error": repr(e)[:200]}
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    B = 32
    # tiny shapes first: correctness of every role incl. ragged tiles / pole rows before the big timings
    run("tiny 64->64 @6x10 B3", 3, 64, 64, 6, 10)
    run("tiny 64->64 @6x10 B3", 3, 64, 64, 6, 10, role="dgrad")
    run("tiny planar 64->128 @8x16 B2", 2, 64, 128, 8, 16, kind="planar")
    run("tiny 128->256 @16x32 B2", 2, 128, 256, 16, 32)
    run("tiny 256->256 @16x32 B2", 2, 256, 256, 16, 32, role="dgrad")
    SHAPES = [("up_3 g|b 128->256 @128x256", 128, 256, 128, 256), ("up_3 g|b 128->128 @128x256", 128, 128, 128, 256),
              ("up_3 conv_0 128->64 @128x256", 128, 64, 128, 256), ("up_3 conv_1 64->64 @128x256", 64, 64, 128, 256),
              ("up_2 g|b 128->512 @64x128", 128, 512, 64, 128), ("up_2 conv_0 256->128 @64x128", 256, 128, 64, 128),
              ("up_1 conv_0 512->256 @32x64", 512, 256, 32, 64), ("up_0 conv_0 1024->512 @16x32", 1024, 512, 16, 32),
              ("G_middle 1024->1024 @8x16", 1024, 1024, 8, 16), ("up_0 g|b 128->2048 @16x32", 128, 2048, 16, 32)]
    for name, C, O, H, W in SHAPES:
        run(name, B, C, O, H, W)
    for name, C, O, H, W in SHAPES[:3] + SHAPES[5:6]:
        run(name, B, C, O, H, W, role="dgrad")
    for name, C, O, H, W in [("vgg 64->64 @128x256", 64, 64, 128, 256), ("vgg 128->128 @64x128", 128, 128, 64, 128),
                             ("vgg 256->256 @32x64", 256, 256, 32, 64), ("vgg 512->512 @16x32", 512, 512, 16, 32)]:
        run(name, B, C, O, H, W, kind="planar")
