#!/bin/bash
# A/B of the CU-partitioned variant of the side-stream weight gradients (hipExtStreamCreateWithCUMask)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs none > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/ab_$tag.json").read().strip().splitlines()[-1])
    print("$tag", "$*", j["value"], "img/s", j["ms_per_step"], "ms")
except Exception as e:
    print("$tag FAILED", e, open("$OUT/ab_$tag.err").read()[-800:])
PY
}
run base EML_WGRAD_OVERLAP=0
run cu4 EML_WGRAD_OVERLAP=1 EML_CU_SPLIT=4 EML_GRID=384 EML_GRID3=192 EML_GRID3_SIDE=64
run cu8 EML_WGRAD_OVERLAP=1 EML_CU_SPLIT=8 EML_GRID=448 EML_GRID3=224 EML_GRID3_SIDE=32
run cu4_ring8 EML_WGRAD_OVERLAP=1 EML_CU_SPLIT=4 EML_GRID=384 EML_GRID3=192 EML_GRID3_SIDE=64 EML_WGRAD_RING=8
# control: the main chain alone on 192 CUs (what the partition costs the HBM-bound kernels)
run g384_only EML_WGRAD_OVERLAP=0 EML_GRID=384 EML_GRID3=192
