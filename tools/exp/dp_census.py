"""Which kernels does the data-parallel path add to a joint iteration?  One profiled iteration, kernel name -> (count, us),
written to gpurun_out/dp_census_<tag>.json; run once plain and once with EML_DIST_SINGLE=1, then `diff` mode prints the difference.
    python tools/exp/dp_census.py run plain | EML_DIST_SINGLE=1 python tools/exp/dp_census.py run single | python tools/exp/dp_census.py diff plain single"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
if sys.argv[1] == "run":
    import torch
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.GenProjector import networks
    from emlight_amd.joint import JointTrainer, joint_batch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    r, local, w = init_distributed()
    torch.manual_seed(0)
    tr = JointTrainer(networks.default_options(), device="cuda:0", world=w)
    batch = joint_batch(32, "cuda:0", seed=9)
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        tr.step(batch)
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            c, t = rows.get(e.name, (0, 0.0))
            rows[e.name] = (c + 1, t + e.device_time_total if hasattr(e, "device_time_total") else t + e.cuda_time_total)
    json.dump(rows, open(os.path.join(out, "dp_census_%s.json" % sys.argv[2]), "w"))
    print(sys.argv[2], len(rows), "kernel names,", sum(c for c, _ in rows.values()), "launches,", round(sum(t for _, t in rows.values()) / 1e3, 2), "ms")
else:
    a = json.load(open(os.path.join(out, "dp_census_%s.json" % sys.argv[2])))
    b = json.load(open(os.path.join(out, "dp_census_%s.json" % sys.argv[3])))
    diff = []
    for k in set(a) | set(b):
        ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
        if ca != cb or abs(tb - ta) > 100:
            diff.append((tb - ta, cb - ca, k))
    for dt, dc, k in sorted(diff, reverse=True)[:25]:
        print("%+9.1f us %+5d launches  %s" % (dt, dc, k[:150]))
    print("...")
    for dt, dc, k in sorted(diff)[:8]:
        print("%+9.1f us %+5d launches  %s" % (dt, dc, k[:150]))
