"""Who launches the ATen glue of a projector / joint step?  One step under torch.profiler; every aten:: op with device time of
its own (GEMMs and convolutions excluded) is attributed to its nearest non-aten ancestor on the CPU side -- the custom
autograd Function (``_SphereConvFnBackward`` ...), the autograd node (``CatBackward0``, ``torch::autograd::AccumulateGrad`` ...)
or, in the forward, the Python call site inside emlight_amd.
    python tools/glue_audit.py [projector|joint] [batch] [vgg: 0|1]"""
import collections
import os
import sys
import warnings

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
warnings.simplefilter("ignore")
which = sys.argv[1] if len(sys.argv) > 1 else "joint"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
vgg = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
from emlight_amd.GenProjector.networks import default_options

opt = default_options(no_vgg_loss=not vgg, vgg_random=True)
if which == "joint":
    from emlight_amd.joint import JointTrainer, joint_batch
    tr = JointTrainer(opt, device="cuda:0")
    data = joint_batch(B, "cuda:0")
else:
    from emlight_amd.GenProjector.data import projector_batch
    from emlight_amd.GenProjector.model_trainer import Trainer
    tr = Trainer(opt, device="cuda:0")
    data = projector_batch(B, "cuda:0")
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()

HEAVY = ("aten::mm", "aten::addmm", "aten::bmm", "aten::convolution", "aten::_convolution", "aten::miopen_convolution",
         "aten::convolution_backward", "aten::_fused_adam_", "aten::_fused_adam")


def dev_us(e):
    return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0) or 0


def owner(e):
    p, chain = e.cpu_parent, []
    while p is not None:
        if not p.name.startswith("aten::"):
            chain.append(p.name.replace("autograd::engine::evaluate_function: ", "")[:48])
            if len(chain) == 2:
                break
        p = p.cpu_parent
    site = ""
    for fr in (e.stack or []):
        if "emlight_amd/" in fr:
            site = fr.split("emlight_amd/")[-1][:60]
            break
    return " <- ".join(chain) + ((" @ " + site) if site else "")


agg = collections.defaultdict(lambda: [0.0, 0])
by_op = collections.defaultdict(lambda: [0.0, 0])
total = 0.0
for e in prof.events():
    t = dev_us(e)
    if not t or not e.name.startswith("aten::") or e.name in HEAVY:
        continue
    k = (e.name, str(e.input_shapes)[:90], owner(e))
    agg[k][0] += t
    agg[k][1] += 1
    by_op[e.name][0] += t
    by_op[e.name][1] += 1
    total += t
print("%s step, B=%d, vgg=%s: ATen glue %.2f ms of device time in %d ops" % (which, B, vgg, total / 1e3, sum(v[1] for v in agg.values())))
print("\n== by op ==")
for k, (t, n) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:30]:
    print("%8.3f ms %5d  %s" % (t / 1e3, n, k))
print("\n== by (op, shapes, owner) ==")
for (name, shapes, own), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:150]:
    print("%8.3f ms %4d  %-26s %-90s %s" % (t / 1e3, n, name, shapes, own))
# launches that are not aten ops (runtime copies / memsets) by their owner
rt = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and ("Memcpy" in e.name or "Memset" in e.name or "copyBuffer" in e.name
                                                                or "fillBuffer" in e.name):
        rt[e.name[:60]][0] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
        rt[e.name[:60]][1] += 1
print("\n== runtime copies / memsets ==")
for k, (t, n) in sorted(rt.items(), key=lambda kv: -kv[1][0])[:10]:
    print("%8.3f ms %5d  %s" % (t / 1e3, n, k))
