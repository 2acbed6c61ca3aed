"""Host-side enqueue time of one projector / joint iteration vs its wall time (is the step launch-bound?)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
which = sys.argv[1] if len(sys.argv) > 1 else "projector"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if which == "joint":
    from emlight_amd.joint import JointTrainer, joint_batch
    from emlight_amd.GenProjector.networks import default_options
    import warnings
    warnings.simplefilter("ignore")
    tr = JointTrainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")
    data = joint_batch(B, "cuda:0")
else:
    from emlight_amd.GenProjector.data import projector_batch
    from emlight_amd.GenProjector.model_trainer import Trainer
    from emlight_amd.GenProjector.networks import default_options
    import warnings
    warnings.simplefilter("ignore")
    tr = Trainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")   # the reference's step: VGG term on
    data = projector_batch(B, "cuda:0")
for _ in range(3):
    tr.step(data)
torch.cuda.synchronize()
host, t0 = 0.0, time.perf_counter()
n = 5
for _ in range(n):
    h0 = time.perf_counter()
    tr.step(data)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("%s B=%d  step %.1f ms  host enqueue %.1f ms per step" % (which, B, tot / n * 1e3, host / n * 1e3))
