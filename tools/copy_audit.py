"""Where do the layout / dtype copies of a projector (or joint) step come from?  One step under torch.profiler with stacks;
aten::copy_ / aten::clone / aten::contiguous device time grouped by input shape and the innermost emlight_amd frame.
    python tools/copy_audit.py [projector|joint] [batch]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "projector"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if which == "joint":
    from emlight_amd.joint import JointTrainer, joint_batch
    tr = JointTrainer(device="cuda:0")
    data = joint_batch(B, "cuda:0")
else:
    from emlight_amd.GenProjector.data import projector_batch
    from emlight_amd.GenProjector.model_trainer import Trainer
    from emlight_amd.GenProjector.networks import default_options
    tr = Trainer(default_options(), device="cuda:0")
    data = projector_batch(B, "cuda:0")
for _ in range(2):
    tr.step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
total = 0.0
for ev in prof.events():
    if ev.name not in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add_", "aten::add", "aten::mul"):
        continue
    t = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    if not t:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "emlight_amd" in fr:
            site = fr.split("emlight_amd/")[-1][:70]
            break
    key = (ev.name, str(ev.input_shapes)[:80], site)
    agg[key][0] += t
    agg[key][1] += 1
    if ev.name == "aten::copy_":
        total += t
print("aten::copy_ device time of one step: %.2f ms (B=%d)" % (total / 1e3, B))
for (name, shapes, site), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.3f ms %4d  %-16s %-80s %s" % (t / 1e3, n, name, shapes, site))
