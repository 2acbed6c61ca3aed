#!/bin/bash
# Same-box A/B of two builds of the library on the regression step: tools/ab_env3.sh <lib A> <lib B>   (EML_LIB_PATH; alternating twice)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for L in "$1" "$2"; do
  ( [ -n "$L" ] && [ "$L" != product ] && export EML_LIB_PATH=$REPO/$L
  timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:22]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('%-28s %7.2f img/s %8.3f ms | %s' % ('$L', j['value'], j['ms_per_step'], {k: v for k, v in f.items() if v}))" )
done
done
