#!/bin/bash
# Same-box A/B of one environment knob on the regression step (families leg), alternating twice: tools/ab_knob_reg.sh KNOB v0 v1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; A=$2; B=$3
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in $A $B; do
  ( export $K=$v
  timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:22]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('$K=$v %7.2f img/s %8.3f ms | %s' % (j['value'], j['ms_per_step'], {k: v for k, v in f.items() if v}))" )
done
done
