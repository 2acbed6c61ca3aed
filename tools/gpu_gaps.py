"""Where the GPU idles inside one projector / joint iteration: device kernels of a profiled step sorted by start time, the idle
gaps between them (> 3 us), summed by the kernel that FOLLOWS the gap -- i.e. which launches the host was late for.
    python tools/gpu_gaps.py [projector|joint] [B]"""
import collections
import os
import sys
import warnings

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
warnings.simplefilter("ignore")
which = sys.argv[1] if len(sys.argv) > 1 else "projector"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
from emlight_amd.GenProjector.networks import default_options  # noqa: E402
if which == "joint":
    from emlight_amd.joint import JointTrainer, joint_batch
    tr = JointTrainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")
    data = joint_batch(B, "cuda:0")
else:
    from emlight_amd.GenProjector.data import projector_batch
    from emlight_amd.GenProjector.model_trainer import Trainer
    tr = Trainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")
    data = projector_batch(B, "cuda:0")
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(data)
    torch.cuda.synchronize()
ks = []
for e in prof.events():
    if str(getattr(e, "device_type", "")).endswith("CUDA") and e.time_range is not None:
        ks.append((e.time_range.start, e.time_range.end, e.name))
ks.sort()
busy = sum(b - a for a, b, _ in ks)
span = ks[-1][1] - ks[0][0]
gaps = collections.defaultdict(lambda: [0, 0.0])
prev_end, prev_name, tot_gap = ks[0][1], ks[0][2], 0.0
big = []
for a, b, n in ks[1:]:
    g = a - prev_end
    if g > 3.0:
        t = gaps[n[:70]]
        t[0] += 1
        t[1] += g
        tot_gap += g
        big.append((g, prev_name[:50], n[:50]))
    prev_end, prev_name = max(prev_end, b), n
print("%s B=%d: %d kernels, span %.1f ms, kernel time %.1f ms, idle in gaps > 3 us: %.1f ms" % (which, B, len(ks), span / 1e3, busy / 1e3, tot_gap / 1e3))
print("(the profiled step's host side is slower than an unprofiled one: an upper bound)")
for n, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%8.2f ms %5d  before %s" % (g / 1e3, c, n))
print("largest single gaps:")
for g, p, n in sorted(big, reverse=True)[:12]:
    print("%8.1f us  after %-50s before %s" % (g, p, n))
