#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for x in 0 1; do
  echo "== EML_WG_XCD=$x"
  EML_WG_XCD=$x timeout 300 python $REPO/tools/bench_wgrad_xcd.py 2>/dev/null
done
done | tee $OUT/r06_wgrad_xcd.txt
