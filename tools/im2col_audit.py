"""Which SphereConv layers of one projector iteration still go through sphere_im2col / sphere_col2im (the 9x operand in HBM),
and what each costs: every C-ABI call of the two entry points timed with events, grouped by (B, source pixels, output pixels, C).
    python tools/im2col_audit.py [B]"""
import collections
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
warnings.simplefilter("ignore")
from emlight_amd import _lib  # noqa: E402
from emlight_amd.GenProjector.data import projector_batch  # noqa: E402
from emlight_amd.GenProjector.model_trainer import Trainer  # noqa: E402
from emlight_amd.GenProjector.networks import default_options  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NARROW = os.environ.get("AUDIT_NARROW", "1") == "1"   # also time the one-pass kernels of the O <= 4 layers
tr = Trainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")
data = projector_batch(B, "cuda:0")
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
real = _lib.lib()
pending = []


class Spy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name not in ("eml_sphere_im2col_f32", "eml_sphere_col2im_f32") and not (NARROW and "narrow" in name and name.endswith("_f32")):
            return fn

        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            ints = tuple(int(v) for v in a if isinstance(v, int))
            ints = ints[-5:-1] if "narrow" in name else ints[:4]          # B, HW, Po, C (the narrow entry points end ..., C, O)
            pending.append((name.replace("eml_sphere_", "").replace("_f32", ""), ints, e0, e1))
            return rc
        return call


_lib.lib = lambda: Spy()
n = 2
for _ in range(n):
    tr.step(data)
torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0, 0.0])
for name, ints, e0, e1 in pending:
    t = tot[(name, ints)]
    t[0] += 1
    t[1] += e0.elapsed_time(e1)
rows = sorted(tot.items(), key=lambda kv: -kv[1][1])
print("%-8s %-34s %6s %9s %9s" % ("pass", "(B, HW, Po, C)", "calls", "ms/iter", "A9 GB"))
s = 0.0
for (name, ints), (cnt, ms) in rows:
    gb = ints[0] * ints[2] * 9 * ints[3] * 4 / 1e9
    s += ms / n
    print("%-8s %-34s %6.1f %9.3f %9.2f" % (name, str(ints), cnt / n, ms / n, gb))
print("total %.2f ms per iteration" % s)
