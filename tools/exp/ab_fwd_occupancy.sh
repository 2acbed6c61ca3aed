#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  env "$@" timeout 300 python $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs families 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={r['kernel'][:18]: r['ms_per_step'] for r in j.get('kernel_families', [])}
print('%-34s %7.2f img/s %8.3f ms | fwd %s' % ('$tag', j['value'], j['ms_per_step'], f.get('conv1x1_fwd_kernel')))"
}
for rep in 1 2; do
run product X=1
run s2_lds_stats EML_LIB_PATH=$REPO/build_exp/lib_s2.so
run s3_3waves_g768_nostage EML_LIB_PATH=$REPO/build_exp/lib_s3.so EML_FWD_STAGED=0 EML_GRID_FWD1=768
run s3_3waves_g512_nostage EML_LIB_PATH=$REPO/build_exp/lib_s3.so EML_FWD_STAGED=0
done
