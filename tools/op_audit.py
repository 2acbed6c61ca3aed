"""Top ATen ops of one projector (or joint) iteration by device time, grouped by input shape (torch.profiler).
    python tools/op_audit.py [projector|joint] [batch]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "projector"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=45,
                                                         max_name_column_width=40, max_shapes_column_width=70))

# library GEMMs: achieved TF/s per shape (find the badly served ones)
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key not in ("aten::mm", "aten::addmm", "aten::bmm") or not ev.input_shapes:
        continue
    sh = [s for s in ev.input_shapes if s]
    try:
        if ev.key == "aten::addmm":
            a, b = sh[1], sh[2]
        else:
            a, b = sh[0], sh[1]
        fl = 2.0 * a[-2] * a[-1] * b[-1] * (a[0] if len(a) == 3 else 1)
    except Exception:
        continue
    t = ev.self_device_time_total / max(ev.count, 1)
    if t > 0:
        rows.append((ev.self_device_time_total / 1e3, ev.count, t / 1e3, fl / t / 1e6, ev.key, a, b))
print("library GEMMs by total device time: total ms, calls, ms/call, TF/s, op, A, B")
for r in sorted(rows, reverse=True)[:30]:
    print("  %7.2f %3d %7.3f %7.1f  %-11s %s x %s" % r)
