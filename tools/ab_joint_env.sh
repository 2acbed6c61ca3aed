#!/bin/bash
# Same-box A/B of environment knobs on the joint step (base / exp / base / exp):   tools/ab_joint_env.sh "<ENV=VAL ...>" [out.txt]
ENVS=$1; OUT=${2:-/dev/stdout}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in base exp base exp; do
  (
  if [ $tag = exp ]; then for kv in $ENVS; do export $kv; done; fi
  timeout 300 python $REPO/bench.py --workload joint --steps 6 --warmup 2 --no_cpu_baseline --legs none 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s [%s] %7.2f img/s %8.3f ms' % ('$tag', '$ENVS' if '$tag'=='exp' else '', j['value'], j['ms_per_step']))"
  )
done >> $OUT
