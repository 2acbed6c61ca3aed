#!/bin/bash
# Same-box A/B of the fused conv3x3 backward: the product (tap-packed weight gradient) against build_exp/lib_nowtp.so
# (tools/exp_build.sh nowtp -DEML_C3_WTP=0: round 4's weight gradient), regression step, alternating, two repetitions.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in old new old new; do
  ( [ $tag = old ] && export EML_LIB_PATH=$REPO/build_exp/lib_nowtp.so
  timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:24]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('%-4s %7.2f img/s %8.3f ms | %s' % ('$tag', j['value'], j['ms_per_step'], {k: v for k, v in f.items() if 'conv3x3' in k}))" )
done
