"""Which autograd / ATen ops of one projector iteration own the non-HIP GPU time (copies, sums, elementwise)?
torch.profiler over 2 iterations, grouped by (op, input shapes) and -- for the glue ops -- by Python call site."""
import os
import sys
import warnings

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
warnings.simplefilter("ignore")
from emlight_amd.GenProjector.data import projector_batch
from emlight_amd.GenProjector.model_trainer import Trainer
from emlight_amd.GenProjector.networks import default_options

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = Trainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")
data = projector_batch(B, "cuda:0")
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
n = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()


def dev_us(e):
    return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)


ev = prof.key_averages(group_by_input_shape=True)
rows = sorted(ev, key=dev_us, reverse=True)
tot = sum(dev_us(e) for e in rows)
print("total device time %.1f ms per iteration" % (tot / n / 1e3))
for e in rows[:70]:
    print("%8.2f ms %5d  %-44s %s" % (dev_us(e) / n / 1e3, e.count // n, e.key[:44], str(e.input_shapes)[:110]))
print("\n== glue ops by input shape ==")
for e in [e for e in rows if e.key.startswith("aten::") and e.key not in ("aten::mm", "aten::addmm", "aten::bmm")][:40]:
    print("%8.2f ms %5d  %-32s %s" % (dev_us(e) / n / 1e3, e.count // n, e.key, str(e.input_shapes)[:120]))
print("\n== glue ops by call site ==")
glue = ("aten::copy_", "aten::sum", "aten::mul", "aten::add", "aten::add_", "aten::clamp_min", "aten::relu", "aten::cat",
        "aten::threshold_backward", "aten::div", "aten::sub", "aten::neg", "aten::fill_", "aten::zero_", "aten::mean",
        "aten::native_batch_norm", "aten::native_batch_norm_backward", "aten::upsample_nearest2d", "aten::clone",
        "aten::contiguous", "aten::leaky_relu", "aten::leaky_relu_backward", "aten::abs", "aten::sgn", "aten::where")
ev2 = prof.key_averages(group_by_stack_n=8)
rows2 = sorted((e for e in ev2 if e.key in glue), key=dev_us, reverse=True)
for e in rows2[:60]:
    site = [s for s in e.stack if "emlight_amd" in s or "bench.py" in s]
    print("%8.2f ms %5d  %-32s %s" % (dev_us(e) / n / 1e3, e.count // n, e.key, " <- ".join(x.split("/")[-1][:60] for x in site[:3])))
