#!/bin/bash
# Same-box sweep of the encoder's grid knobs on the regression step (families leg): tools/exp/grid_sweep.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  ( [ -n "$1" ] && export $1=$2
  timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:14]: round(r['ms_per_step'],2) for r in j.get('kernel_families', [])}
print('%-22s %7.2f img/s %8.3f ms | %s' % ('$1=$2', j['value'], j['ms_per_step'], {k: v for k, v in f.items() if v}))" )
}
run "" ""
for k in EML_GRID_FWD1 EML_GRID_WGRAD1 EML_GRID_DGRAD; do for v in 256 768 1024; do run $k $v; done; done
run "" ""
for v in 128 512; do run EML_GRID3 $v; run EML_GRID3_DGRAD $v; done
run "" ""
