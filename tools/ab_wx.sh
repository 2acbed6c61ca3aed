#!/bin/bash
# experiment builds of conv3x3_bwd_weight_kernel (-DEML_WX bits) against the product, separate launches (EML_C3_FOLD=0)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export EML_C3_FOLD=0
for tag in head wx8 wx9 head; do
  (
  [ $tag != head ] && export EML_LIB_PATH=$REPO/build_exp/lib_$tag.so
  timeout 300 python $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'].split(' ')[0]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('%-5s %8.3f ms | 3x3_bwd_weight %.2f  3x3_bwd_data %.2f  3x3_fwd %.2f' % ('$tag', j['ms_per_step'], f.get('conv3x3_bwd_weight_kernel',-1), f.get('conv3x3_bwd_data_kernel',-1), f.get('conv3x3_fwd_kernel',-1)))"
  )
done
