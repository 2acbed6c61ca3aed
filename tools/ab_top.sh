#!/bin/bash
# Same-box A/B of the round's two line-granularity fixes, regression step, alternating, two repetitions:
#   EML_DGRAD_TOP=0/1  the two-layer data-gradient pass's top 24 columns as compact tensors (their readers fetch 48 bytes per
#                      pixel instead of one or two 128-byte lines of a wide G row)
#   EML_ROW_ALIGN=16/32  block-buffer rows of whole 128-byte lines (block 2: 304 -> 320 floats)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "0 16" "1 16" "1 32" "0 32"; do
  set -- $cfg
  ( export EML_DGRAD_TOP=$1 EML_ROW_ALIGN=$2
  timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:22]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('top=$1 align=$2 %7.2f img/s %8.3f ms | %s' % (j['value'], j['ms_per_step'], {k: v for k, v in f.items() if v}))" )
done
done
