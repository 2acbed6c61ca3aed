#!/bin/bash
# round-5 GPU call I: split-K target of the fused weight gradient (default now 1024): 512 / 768 / 1536 against it
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
rm -f $OUT/r05i_ab.txt
for v in 512 768 1536; do bash tools/ab_joint_env.sh "EML_WGRAD_WGS=$v" $OUT/r05i_ab.txt; done
cat $OUT/r05i_ab.txt
