#!/bin/bash
# Same-box A/B of one environment knob on the projector and joint legs, alternating, two repetitions:
#   tools/ab_knob.sh EML_NARROW_PROJECT 0 1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; A=$2; B=$3
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in $A $B; do
  ( export $K=$v
  timeout 600 python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs projector,joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$K=$v  regression %7.2f img/s %8.3f ms | projector %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % (j['value'], j['ms_per_step'], j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['step_frac_of_f32_mfma_peak']))" )
done
done
