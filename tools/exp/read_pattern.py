"""(build first: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/exp/read_pattern.hip -o build_exp/librp.so)
Read-pattern probe (tools/exp/read_pattern.hip, build_exp/librp.so): TB/s of reading the first `cols` columns of a (P, LD) f32 matrix
in the access shapes the 1x1 kernels use against whole lines / whole rows."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_exp", "librp.so"))
lib.rp_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
out = torch.empty(4096 * 256, device="cuda")
for P, LD, nc in [(4915200, 224, 192), (4915200, 224, 96), (4915200, 224, 32), (4915200, 48, 48)]:
    G = torch.randn(P * LD, device="cuda")
    res = []
    for which, name in ((0, "D^T 16 rows x 64 B"), (1, "8 lanes/row 128 B"), (2, "rows"), (3, "float2, 4 rows x 128 B")):
        if which == 3 and nc % 32: continue
        for grid in (512, 2048):
            ms = t(lambda: lib.rp_run(which, G.data_ptr(), P, LD, nc, grid, out.data_ptr(), st))
            res.append("%s g%d %.2f" % (name, grid, P * nc * 4 / ms / 1e9))
    print("P=%d LD=%d cols=%d (TB/s of the columns asked for): " % (P, LD, nc) + " | ".join(res))
