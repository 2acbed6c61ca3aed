"""Host side of tools/exp/row_access_probe.hip: GB/s of the two access shapes on a block-1-sized buffer."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "build_exp", "row_probe.so"))
lib.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_void_p, ctypes.c_void_p]
P, ld = 64 * 240 * 320, 224
X = torch.zeros(P, ld, device="cuda")
sink = torch.zeros(4, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(mode, k, grid, reps=5):
    f = lambda: lib.probe_launch(ctypes.c_void_p(X.data_ptr()), ld, P, k, mode, grid, ctypes.c_void_p(sink.data_ptr()), st)
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = P * k * 4 * (2 if mode in (2, 3) else 1)
    return ms, nbytes / ms / 1e6


names = {0: "read, operand shape (16 rows x 64 B per wave instruction)", 1: "read, whole rows (one row per wave instruction)",
         2: "read-modify-write, operand shape", 3: "read-modify-write, whole rows",
         4: "read, 1x1 weight gradient's x operand (4 rows x 128 B, 8 B per lane)",
         5: "read, the same as float4 (4 rows x 256 B, 16 B per lane)"}
for k in (64, 128, 208, 224):
    for mode in ((4, 5) if os.environ.get("PROBE_WGRAD_ONLY") else (0, 1, 2, 3, 4, 5)):
        for grid in (512, 1024):
            ms, gbps = run(mode, k, grid)
            print("k=%3d grid=%4d  %-62s %7.3f ms  %7.1f GB/s" % (k, grid, names[mode], ms, gbps), flush=True)
