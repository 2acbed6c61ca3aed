"""Which host <-> device copies and synchronisations does ONE steady-state joint iteration still make?  (Round 6: the stream
capture probe found a Python scalar stored into a device tensor -- a pageable host-to-device copy, i.e. a host synchronisation --
46 times per projector iteration.)  One iteration under torch.profiler: every runtime call that copies or waits, with the
Python frame that issued it.     python tools/host_syncs.py [B]"""
import collections
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
warnings.simplefilter("ignore")
from emlight_amd import _runtime  # noqa: E402
_runtime.entry_point_defaults()
from emlight_amd.GenProjector.networks import default_options  # noqa: E402
from emlight_amd.joint import JointTrainer, joint_batch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tr = JointTrainer(default_options(no_vgg_loss=False, vgg_random=True), device="cuda:0")
batch = joint_batch(B, "cuda:0")
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=None) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
calls = collections.Counter()
where = collections.defaultdict(collections.Counter)
dev_copies, sync_ops = collections.Counter(), collections.defaultdict(collections.Counter)
for e in prof.events():
    n = e.name
    if n.startswith("Memcpy") or n.startswith("Memset"):
        dev_copies[n] += 1
    if n in ("aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::copy_", "aten::to", "aten::_to_copy", "aten::fill_"):
        st = [f.strip() for f in (e.stack or [])]
        mine = [f for f in st if "emlight_amd" in f or "bench.py" in f]
        if n in ("aten::item", "aten::_local_scalar_dense", "aten::is_nonzero"):
            sync_ops[n][mine[0][:140] if mine else (st[0][:100] if st else "?")] += 1
    if not (n.startswith("hip") or n.startswith("cuda")):
        continue
    if "LaunchKernel" in n or "GetLastError" in n or "hipGetDevice" in n or "PeekAtLastError" in n or "hipSetDevice" in n:
        continue
    calls[n] += 1
    if "Memcpy" in n or "Synchronize" in n or "Memset" in n or "EventQuery" in n:
        frames = [f for f in (e.stack or []) if "emlight_amd" in f or "bench" in f or "joint" in f]
        st = [f.strip()[:90] for f in (e.stack or [])][:4]
        where[n][frames[0].strip()[:150] if frames else "(no frame of this repo: %s)" % " <- ".join(st)] += 1
print("device-side copies / memsets of one joint iteration:", dict(dev_copies))
for n, c in sync_ops.items():
    print("host-synchronising op %s:" % n)
    for w, k in c.most_common(10):
        print("     %4d  %s" % (k, w))
print("runtime calls of one joint iteration (B = %d) other than kernel launches:" % B)
for n, c in calls.most_common():
    print("  %5d  %s" % (c, n))
    for w, k in where[n].most_common(8):
        print("         %4d  %s" % (k, w))
