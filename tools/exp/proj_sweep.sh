#!/bin/bash
# Same-box sweep of projector-side knobs on the projector and joint legs: tools/exp/proj_sweep.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  ( [ -n "$1" ] && export $1=$2
  timeout 600 python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs projector,joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s projector %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % ('$1=$2', j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['step_frac_of_f32_mfma_peak']))" )
}
run "" ""
for v in 512 768 1536; do run EML_WGRAD_WGS $v; done
run "" ""
for v in 32 128; do run EML_FUSED_MIN_MB $v; done
run EML_WGRAD_LIB_KEPT 0
run "" ""
