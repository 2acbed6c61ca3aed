#!/bin/bash
# Weak-scaling curve on an N-GPU node (not runnable on the 1-GPU gpurun box; for the day a node exists):
#   tools/scale_check.sh [legs] [gpu counts...]     e.g.  tools/scale_check.sh joint 1 2 4 8
# Runs `bench.py --gpus N` (one process per GPU over RCCL/xGMI; bench.py starts its own ranks) for each N and prints the
# whole-job img/s, the per-GPU img/s and the efficiency against N x the 1-GPU value, for the headline regression step and
# for every leg object asked for.
REPO=$(cd "$(dirname "$0")/.." && pwd)
# `tools/scale_check.sh dry`: no node needed -- two ranks on ONE GPU (gloo transport) run a projector iteration and the test
# asserts what it puts on the wire besides the gradient buckets: 14 sync-BN all-reduces per generator forward, 18 per backward,
# 14 for the discriminator step's no-grad generator pass, (2C+1) f64 each; no D buckets in the G step, no G buckets in the D step
if [ "$1" = dry ]; then
  cd $REPO && exec python -m pytest tests/test_gpu_ddp_two_ranks.py -q -m gpu -k "collective_counts or sync_bn_two_ranks"
fi
# `tools/scale_check.sh rccl`: no node needed either -- the FIRST RCCL run of this code need not be the 8-GPU one: one rank on the
# one GPU with EML_DIST_SINGLE=1 (emlight_amd/_dist.py) initialises the "nccl" process group, wraps the three networks in DDP
# and issues every collective of an iteration through RCCL; then the bench's joint leg the same way, which prints what it saw
# (`collectives`: ranks, explicit all-reduces, DDP buckets).  (Two ranks on one device: RCCL answers "Duplicate GPU detected".)
if [ "$1" = rccl ]; then
  cd $REPO && python -m pytest tests/test_gpu_ddp_two_ranks.py -q -m gpu -k "one_rank_rccl" || exit 1
  EML_DIST_SINGLE=1 exec python bench.py --gpus 1 --steps 5 --warmup 2 --no_cpu_baseline --legs joint
fi
LEGS=${1:-joint}
shift
NS=${@:-1 2 4 8}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
OUT=${OUT:-$REPO/gpurun_out}
mkdir -p $OUT
for N in $NS; do
  timeout 1800 python $REPO/bench.py --gpus $N --steps 10 --warmup 3 --no_cpu_baseline --legs $LEGS > $OUT/scale_$N.json 2> $OUT/scale_$N.err \
    || { echo "N=$N failed:"; tail -5 $OUT/scale_$N.err; }
done
python - "$OUT" $NS <<'PY'
import json, sys
out, ns = sys.argv[1], [int(n) for n in sys.argv[2:]]
rows = {}
for n in ns:
    try:
        j = json.loads(open("%s/scale_%d.json" % (out, n)).read().strip().splitlines()[-1])
    except Exception as e:
        print("N=%d: no line (%s)" % (n, e))
        continue
    rows[n] = {"regression": j["value"], **{k: j[k]["value"] for k in ("projector", "joint") if k in j}}
base = rows.get(ns[0])
for leg in (base or {}):
    print(leg)
    for n, r in rows.items():
        if leg in r:
            print("  N=%d  %9.2f img/s  %8.2f per GPU  efficiency %.3f" % (n, r[leg], r[leg] / n, r[leg] / (n / ns[0] * base[leg])))
PY
