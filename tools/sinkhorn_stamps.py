"""Phase timestamps of the register-resident Sinkhorn kernel (an experiment build, not the product library):
    tools/exp_build.sh stamps -DEML_STAMPS
    EML_LIB_PATH=build_exp/lib_stamps.so python tools/sinkhorn_stamps.py [B N blur]
Prints, for three loss calls, the time of each phase boundary in ns since kernel start for sample 0's coupled workgroup
(DESIGN.md section 3.3 tabulates one such run)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from emlight_amd import _lib  # noqa: E402

B, N, blur = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (64, 128, .05)
L = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(L, "eml_sinkhorn_read_stamps"):
    raise SystemExit("this library was not built with -DEML_STAMPS (see the docstring)")
names = ["start", "own-staging", "schedule+staged", "costs"]
buf = (ctypes.c_longlong * 64)()
for rep in range(3):
    r = bench.time_sinkhorn(B, N, blur, "cuda:0", reps=20 + rep)
    torch.cuda.synchronize()
    L.eml_sinkhorn_read_stamps(buf)
    v = list(buf)
    n = 4 + r["sweeps"]
    t = [(v[i] - v[0]) * 10 for i in range(n)]
    print("%.1f us per loss call (with stamps) |" % (r["ms_per_loss_call"] * 1e3),
          "  ".join("%s %d" % (names[i] if i < 4 else "sweep%d" % (i - 4), t[i]) for i in range(n)), "ns")
