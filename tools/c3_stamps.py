"""Where a tile of the fused conv3x3 backward spends its cycles (an experiment build, not the product library):
    tools/exp_build.sh stamps -DEML_STAMPS
    EML_LIB_PATH=build_exp/lib_stamps.so python tools/c3_stamps.py
Runs 3 regression steps and prints the shader-clock cycles per tile wave 0 of a workgroup spends in phase A (data gradient),
at the barrier after it, in phase B (weight gradient + the data gradient's epilogue) and at the closing barrier."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emlight_amd import _lib  # noqa: E402
from emlight_amd.RegressionNetwork.data import synthetic_batch  # noqa: E402
from emlight_amd.RegressionNetwork.engine import RegressionTrainer  # noqa: E402

L = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(L, "eml_c3_read_stamps"):
    raise SystemExit("this library was not built with -DEML_STAMPS (see the docstring)")
tr = RegressionTrainer(anchors=128, crop_hw=(240, 320), blur=.05, device="cuda:0")
batch = synthetic_batch(64, 128, (240, 320), seed=1, device="cuda:0")
buf = (ctypes.c_ulonglong * 8)()
for rep in range(3):
    tr.step(batch)
    torch.cuda.synchronize()
    L.eml_c3_read_stamps(buf, 1)
    v = list(buf)
    n = max(v[4], 1)
    tot = sum(v[:4])
    print("tiles %d (wave 0 of every workgroup): per tile  phase A %.0f  barrier2 %.0f  phase B %.0f  barrier1 %.0f  = %.0f cycles"
          % (v[4], v[0] / n, v[1] / n, v[2] / n, v[3] / n, tot / n), "| MFMA issue time of a wave pair: %d" % (2 * 378 * 32))
