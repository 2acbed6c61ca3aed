#!/bin/bash
# Same-box A/B of a previous build of the library (build_exp/lib_<name>.so via EML_LIB_PATH) against the in-tree one on the
# projector step (prev / head / prev / head):   tools/ab_projector_lib.sh <name> [out.txt]
NAME=$1; OUT=${2:-/dev/stdout}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in prev head prev head; do
  (
  if [ $tag = prev ]; then export EML_LIB_PATH=$REPO/build_exp/lib_$NAME.so; fi
  timeout 300 python $REPO/bench.py --workload projector --steps 8 --warmup 3 --no_cpu_baseline --legs none 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s %7.2f img/s %8.3f ms' % ('$tag', j['value'], j['ms_per_step']))"
  )
done >> $OUT
