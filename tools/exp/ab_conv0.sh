#!/bin/bash
# Same-box A/B of conv0's forward on the matrix unit (EML_CONV0_MFMA=1, default) against the VALU kernel (0): tools/exp/ab_conv0.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1; do
  ( export EML_CONV0_MFMA=$v
  timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EML_CONV0_MFMA=$v %7.2f img/s %8.3f ms' % (j['value'], j['ms_per_step']))" )
done; done
