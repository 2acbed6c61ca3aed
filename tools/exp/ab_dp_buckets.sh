#!/bin/bash
# Same-box A/B of the gradient reducer on the one-rank RCCL dry run (EML_DIST_SINGLE=1): EML_DP_BUCKETS=1 (GradientBuckets)
# against 0 (torch's DistributedDataParallel), and the plain single-process step beside them; alternating twice.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  ( export EML_DIST_SINGLE=$1 EML_DP_BUCKETS=$2
  timeout 600 python $REPO/bench.py --gpus 1 --steps 8 --warmup 3 --no_cpu_baseline --legs joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=(j['joint'].get('collectives') or {}).get('ddp')
ob={b['bucket'][:12]: (b.get('launches_per_step'), b['ms_per_step']) for b in j['joint'].get('other_breakdown', [])}
print('   ', ob)
print('dist_single=$1 buckets=$2  regression %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms | %s' % (j['value'], j['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], {k: (v.get('buckets'), v.get('bytes')) for k, v in (c or {}).items()}))" )
}
for rep in 1 2; do run 0 1; run 1 1; run 1 0; done
