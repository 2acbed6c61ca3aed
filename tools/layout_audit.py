"""Which .contiguous() calls of a projector step actually copy (layout conversions)?  Prints call sites by bytes."""
import sys, os, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd.GenProjector.networks import default_options
from emlight_amd.GenProjector.model_trainer import Trainer
from emlight_amd.GenProjector.data import projector_batch
tr = Trainer(default_options(), device="cuda")
data = projector_batch(4, "cuda")
tr.step(data)
orig = torch.Tensor.contiguous
log = collections.Counter()
def patched(self, *a, **k):
    out = orig(self, *a, **k)
    if out.data_ptr() != self.data_ptr() and self.numel() > 1 << 16:
        fr = [f for f in traceback.extract_stack()[:-1] if "emlight_amd" in f.filename][-1:]
        site = "%s:%d" % (os.path.basename(fr[0].filename), fr[0].lineno) if fr else "?"
        log[(site, tuple(self.shape), tuple(self.stride()))] += self.numel() * 4
    return out
torch.Tensor.contiguous = patched
tr.step(data)
torch.cuda.synchronize()
for (site, shape, stride), b in log.most_common(25):
    print("%8.1f MB  %-28s %s %s" % (b / 1e6, site, shape, stride))
