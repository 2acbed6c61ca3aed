"""(build first: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/exp/write_pattern.hip -o build_exp/libwp.so)
Write-pattern probe (tools/exp/write_pattern.hip, build_exp/libwp.so): TB/s of writing a (P, LD) f32 matrix in the D^T epilogue
pattern (16 rows x 64 B per instruction), row by row (whole lines), and flat."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build_exp", "libwp.so"))
lib.wp_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for P, LD, nc in [(2457600, 224, 224), (2457600, 224, 192), (2457600, 224, 96), (1048576, 128, 128), (4915200, 48, 48), (4915200, 224, 32)]:
    G = torch.empty(P * LD, device="cuda")
    out = []
    for which, name in ((0, "D^T lanes (r,kk)"), (3, "4 lanes/row 64B"), (4, "8 lanes/row 128B"), (1, "rows"), (2, "flat")):
        for grid in (512,):
            ms = t(lambda: lib.wp_run(which, G.data_ptr(), P, LD, nc, grid, st))
            nbytes = P * (LD if which == 2 else nc) * 4
            out.append("%s g%d %.2f TB/s" % (name, grid, nbytes / ms / 1e9))
    print("P=%d LD=%d cols=%d: " % (P, LD, nc) + " | ".join(out))
