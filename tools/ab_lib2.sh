#!/bin/bash
# Same-box A/B of a previous build of the library (build_exp/lib_<name>.so, via EML_LIB_PATH) against the in-tree one on the
# regression step's kernel families (exp / base / exp / base):   tools/ab_lib2.sh <name> [out.txt]
NAME=$1; OUT=${2:-/dev/stdout}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in prev head prev head; do
  (
  if [ $tag = prev ]; then export EML_LIB_PATH=$REPO/build_exp/lib_$NAME.so; fi
  timeout 300 python $REPO/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'].split(' ')[0]: r['ms_per_step'] for r in j.get('kernel_families', [])}
keys=['conv1x1_fwd_kernel','conv1x1_bwd_weight_kernel','conv1x1_bwd_data_multi_kernel','conv3x3_fwd_kernel','conv3x3_bwd_fused_kernel','conv3x3_bwd_data_kernel','conv3x3_bwd_weight_kernel','transition_bwd_data_kernel']
print('%-5s %7.2f img/s %8.3f ms | ' % ('$tag', j['value'], j['ms_per_step']) + ' '.join('%s %.2f' % (k.replace('_kernel','').replace('conv',''), f.get(k, -1)) for k in keys))"
  )
done >> $OUT
