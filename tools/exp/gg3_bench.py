"""Round 6: what each piece of the footprint gather-GEMM's loop costs.  Times eml_sphere_conv_lowres_f32 of every
build_exp/libgg3_*.so (tools/exp/gg3_variants.sh) on a few forward shapes, next to the library GEMM on the materialised operand
(with and without its im2col pass).    python tools/exp/gg3_bench.py [reps]"""
import ctypes
import glob
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from emlight_amd import _lib, _runtime  # noqa: E402
_runtime.entry_point_defaults()
from emlight_amd.GenProjector.spherenet import _lowres_plan, _SphereConvFn, sphere_geometry  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ONLY = os.environ.get("GG3_ONLY")
vp = ctypes.c_void_p
LIBS = {}
for path in sorted(glob.glob(os.path.join(ROOT, "build_exp", "libgg3_*.so"))):
    h = ctypes.CDLL(path)
    h.eml_sphere_conv_lowres_f32.restype = ctypes.c_int
    h.eml_sphere_conv_lowres_f32.argtypes = [vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp] + [ctypes.c_int] * 6 + \
        [vp, ctypes.c_float, vp]
    LIBS[os.path.basename(path)[len("libgg3_"):-3]] = h
L, p = _lib.lib(), _lib.ptr
LIBS = {"shipped": L, **LIBS}
if os.environ.get("GG3_VARIANT"):      # PMC passes: one variant per process
    LIBS = {k: v for k, v in LIBS.items() if k == os.environ["GG3_VARIANT"]}
FWD_ONLY = bool(os.environ.get("GG3_VARIANT"))


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


def run(name, B, C, O, H, W, split_override=None):
    if ONLY and ONLY not in name:
        return
    dev = torch.device("cuda")
    geo = sphere_geometry(H, W, 1, dev)
    po = H * W
    torch.manual_seed(1)
    x = torch.randn(B, H, W, C, device=dev)
    w2 = torch.randn(O, 9 * C, device=dev) * 0.05
    bias = torch.randn(O, device=dev)
    fp, fp_max, split, lidx = _lowres_plan(geo, B, C, O)
    if split_override:
        split = split_override
    y = torch.empty(B * po, O, device=dev)
    part = torch.empty(max(1, split * B * po * O if split > 1 else 1), device=dev)
    st = _lib.current_stream()
    gflop = 2.0 * B * po * 9 * C * O / 1e9
    row = {"layer": name, "B": B, "gflop": round(gflop, 1), "split": split, "fp_max": fp_max}
    for vname, h in LIBS.items():
        def call():
            rc = h.eml_sphere_conv_lowres_f32(p(x), p(lidx), p(geo.wgt), None, 4, p(fp), fp_max, p(w2), p(bias), p(y), p(part), split,
                                              B, po, po, C, O, None, 1.0, st)
            assert rc == 0, rc
        ms = events(call)
        row[vname] = {"ms": round(ms, 4), "tflops": round(gflop / ms, 1)}
    if FWD_ONLY:
        print(json.dumps(row), flush=True)
        return
    a9 = _SphereConvFn._im2col(x, geo, B, C)
    ms_g = events(lambda: torch.addmm(bias, a9, w2.t()))
    ms_i = events(lambda: _SphereConvFn._im2col(x, geo, B, C))
    row["library"] = {"gemm_ms": round(ms_g, 4), "gemm_tflops": round(gflop / ms_g, 1), "im2col_ms": round(ms_i, 4),
                      "total_tflops": round(gflop / (ms_g + ms_i), 1)}
    print(json.dumps(row), flush=True)


run("G_middle 1024->1024 @8x16", 32, 1024, 1024, 8, 16)


run("up_0 1024->512 @16x32", 32, 1024, 512, 16, 32)
run("up_0 g|b 128->2048 @16x32", 32, 128, 2048, 16, 32)
run("up_1 512->256 @32x64", 32, 512, 256, 32, 64)
run("up_1 g|b 128->1024 @32x64", 32, 128, 1024, 32, 64)


def run_bwd(name, B, C, O, H, W):
    """The two halves of the backward of one layer, each way: input gradient = the footprint kernel over the transposed table
    vs dY W2 (library) + sphere_col2im; weight gradient = library GEMM on the kept operand vs im2col rebuilt + library GEMM vs the
    fused split-K kernel."""
    if (ONLY and ONLY not in name) or FWD_ONLY:
        return
    dev = torch.device("cuda")
    geo = sphere_geometry(H, W, 1, dev)
    po = H * W
    torch.manual_seed(2)
    x = torch.randn(B, H, W, C, device=dev)
    gy = torch.randn(B * po, O, device=dev)
    w2 = torch.randn(O, 9 * C, device=dev) * 0.05
    w2t = w2.view(O, 9, C).permute(2, 1, 0).reshape(C, 9 * O).contiguous()
    st = _lib.current_stream()
    gflop = 2.0 * B * po * 9 * C * O / 1e9
    row = {"layer": name, "role": "backward", "B": B, "gflop": round(gflop, 1)}
    plan = _lowres_plan(geo, B, O, C, transposed=True)
    tidx, twgt, rowmax, ke = geo.transposed_table()
    if plan is not None:
        fp, fp_max, split, tlidx = plan
        gx = torch.empty(B * po, C, device=dev)
        part = torch.empty(max(1, split * B * po * C if split > 1 else 1), device=dev)

        def dgrad3():
            rc = L.eml_sphere_conv_lowres_f32(p(gy), p(tlidx), p(twgt), p(rowmax) if ke == 8 else None, ke, p(fp), fp_max, p(w2t), None, p(gx),
                                              p(part), split, B, po, po, O, C, None, 1.0, st)
            assert rc == 0
        ms = events(dgrad3)
        row["dgrad_footprint"] = {"ms": round(ms, 4), "tflops": round(gflop / ms, 1), "split": split, "ke": ke, "fp_max": fp_max}
    gxr = torch.empty(B, H, W, C, device=dev)

    def dgrad_lib():
        da9 = gy @ w2
        L.eml_sphere_col2im_f32(p(da9), p(geo.csr_ptr), p(geo.csr_src), p(geo.csr_w), p(gxr), B, po, po, C, st)
    ms = events(dgrad_lib)
    ms_mm = events(lambda: gy @ w2)
    row["dgrad_library"] = {"ms": round(ms, 4), "tflops": round(gflop / ms, 1), "gemm_ms": round(ms_mm, 4)}
    a9 = _SphereConvFn._im2col(x, geo, B, C)
    ms_k = events(lambda: (a9.t() @ gy).t().contiguous())
    ms_i = events(lambda: _SphereConvFn._im2col(x, geo, B, C))
    row["wgrad_library_kept"] = {"ms": round(ms_k, 4), "tflops": round(gflop / ms_k, 1), "im2col_ms": round(ms_i, 4)}
    del a9
    if C % 64 == 0:
        from emlight_amd.GenProjector.spherenet import SphereConv2D
        bn = 128 if C % 128 == 0 else 64
        bmo = 128 if (O % 128 == 0 or O > 192) else 64
        tiles = 9 * (C // bn) * ((O + bmo - 1) // bmo)
        for wgs in (512, 1024, 2048):
            split = max(1, min((B * po + 31) // 32, wgs // tiles))
            partw = torch.empty(L.eml_sphere_conv_wgrad_partial_floats(C, O, split), device=dev)
            gw2 = torch.empty(O, 9 * C, device=dev)
            ms = events(lambda: L.eml_sphere_conv_wgrad_fused_f32(p(x), p(geo.idx), p(geo.wgt), p(gy), p(partw), p(gw2), B, po, po, C, O, split, st))
            row["wgrad_fused_%d" % wgs] = {"ms": round(ms, 4), "tflops": round(gflop / ms, 1), "split": split, "tiles": tiles}
    print(json.dumps(row), flush=True)


run_bwd("G_middle 1024->1024 @8x16", 32, 1024, 1024, 8, 16)
run_bwd("up_0 1024->512 @16x32", 32, 1024, 512, 16, 32)
run_bwd("up_0 512->512 @16x32", 32, 512, 512, 16, 32)
run_bwd("up_0 g|b 128->2048 @16x32", 32, 128, 2048, 16, 32)
run_bwd("up_1 512->256 @32x64", 32, 512, 256, 32, 64)
run_bwd("up_1 g|b 128->1024 @32x64", 32, 128, 1024, 32, 64)
run_bwd("D 256->512 @16x32", 64, 256, 512, 16, 32)
