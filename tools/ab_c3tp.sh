#!/bin/bash
# Same-box A/B of the tap-packed conv3x3 forward (EML_C3_TP=auto, the default) against the halo-tile kernel everywhere (off):
# regression step, alternating, two repetitions.  tools/ab_c3tp.sh [legs]
LEGS=${1:-families}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in off auto off auto; do
  EML_C3_TP=$tag timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs $LEGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:24]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('EML_C3_TP=%-5s %7.2f img/s %8.3f ms | %s' % ('$tag', j['value'], j['ms_per_step'], {k: v for k, v in f.items() if 'conv3x3' in k}))"
done
