#!/bin/bash
# A/B of an experiment build + environment knobs against the product: tools/ab_env.sh <lib name|-> "<ENV=VAL ...>" [legs]
NAME=$1; ENVS=$2; LEGS=${3:-families}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in base exp base exp; do
  (
  if [ $tag = exp ]; then
    [ "$NAME" != "-" ] && export EML_LIB_PATH=$REPO/build_exp/lib_$NAME.so
    for kv in $ENVS; do export $kv; done
  fi
  timeout 300 python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs $LEGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'].split(' ')[0]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('%-5s %7.2f img/s %8.3f ms | fwd %s wgrad %s dgrad %s' % ('$tag', j['value'], j['ms_per_step'], f.get('conv1x1_fwd_kernel'), f.get('conv1x1_bwd_weight_kernel'), f.get('conv1x1_bwd_data_multi_kernel')))"
  )
done
