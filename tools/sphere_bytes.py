"""A9 bytes and time of the sphere gathers in one projector step (B from argv, default 32)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd.GenProjector import spherenet
from emlight_amd.GenProjector.networks import default_options
from emlight_amd.GenProjector.model_trainer import Trainer
from emlight_amd.GenProjector.data import projector_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = Trainer(default_options(), device="cuda")
data = projector_batch(B, "cuda")
tr.step(data)
stats = collections.defaultdict(lambda: [0, 0.0, 0])
orig = spherenet._SphereConvFn._im2col
def timed(xr, geo, b, c):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(xr, geo, b, c); e1.record()
    stats[(geo.ho, geo.wo, c)][0] += out.numel() * 4
    stats[(geo.ho, geo.wo, c)][2] += 1
    ev.append(((geo.ho, geo.wo, c), e0, e1))
    return out
ev = []
spherenet._SphereConvFn._im2col = staticmethod(timed)
tr.step(data)
torch.cuda.synchronize()
for k, a, b in ev:
    stats[k][1] += a.elapsed_time(b)
tot_b = tot_t = 0
for k, (by, ms, n) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print("Ho,Wo,C=%-16s calls %3d  A9 %7.2f GB  %7.2f ms  %6.2f TB/s written" % (k, n, by / 1e9, ms, by / ms / 1e9 if ms else 0))
    tot_b += by; tot_t += ms
print("total A9 %.1f GB in %.1f ms" % (tot_b / 1e9, tot_t))
