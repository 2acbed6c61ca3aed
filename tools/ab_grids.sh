#!/bin/bash
# per-family persistent-grid sweep of the 1x1 kernels (one gpurun call): tools/ab_grids.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs families > $OUT/abg_$tag.json 2> $OUT/abg_$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/abg_$tag.json").read().strip().splitlines()[-1])
    f = {r["kernel"].split(" ")[0]: r["ms_per_step"] for r in j["kernel_families"]}
    print("%-10s %-40s %7.2f img/s %8.3f ms | fwd %.2f wgrad %.2f dgrad %.2f" % ("$tag", "$*", j["value"], j["ms_per_step"],
          f["conv1x1_fwd_kernel"], f["conv1x1_bwd_weight_kernel"], f["conv1x1_bwd_data_multi_kernel"]))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/abg_$tag.err").read()[-500:])
PY
}
run base X=1
run d768 EML_GRID_DGRAD=768
run d1024 EML_GRID_DGRAD=1024
run d256 EML_GRID_DGRAD=256
run f768 EML_GRID_FWD1=768
run f1024 EML_GRID_FWD1=1024
run f256 EML_GRID_FWD1=256
run w256 EML_GRID_WGRAD1=256
run w768 EML_GRID_WGRAD1=768
run base2 X=1
