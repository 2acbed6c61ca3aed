#!/bin/bash
# A/B of the side-stream conv3x3 weight gradients (one gpurun call):  tools/ab_overlap.sh
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
run ov EML_WGRAD_OVERLAP=1
run ov_s128 EML_WGRAD_OVERLAP=1 EML_GRID3_SIDE=128
run ov_s64_g384 EML_WGRAD_OVERLAP=1 EML_GRID3_SIDE=64 EML_GRID=384
run ov_s64 EML_WGRAD_OVERLAP=1 EML_GRID3_SIDE=64
run ov_lo EML_WGRAD_OVERLAP=1 EML_SIDE_PRIO=1
run ov_hi EML_WGRAD_OVERLAP=1 EML_SIDE_PRIO=-1
run base2 EML_WGRAD_OVERLAP=0
EML_WGRAD_OVERLAP=1 timeout 600 python -m pytest $REPO/tests/test_gpu_densenet.py -x -q 2>&1 | tail -3
rm -rf /tmp/ovt
EML_WGRAD_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovt -o k -- python $REPO/bench.py --steps 2 --warmup 1 --no_cpu_baseline --legs none > $OUT/ab_trace.log 2>&1
python $REPO/tools/overlap_trace.py /tmp/ovt | tee $OUT/ab_overlap_trace.txt
