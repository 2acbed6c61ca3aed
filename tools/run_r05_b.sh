#!/bin/bash
# round-5 GPU call B: parity of the fused L1 terms / the small-layer input gradient (whole -m gpu suite), then the same-box A/B
# of the two new paths on the joint and projector steps ("exp" = knobs set = the round-4 dispatch)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r05b_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05b_pytest.txt
tail -4 $OUT/r05b_pytest.txt
rm -f $OUT/r05b_ab.txt
bash tools/ab_joint_env.sh "EML_FUSED_L1=0 EML_SMALL_DA9=0" $OUT/r05b_ab.txt
bash tools/ab_joint_env.sh "EML_FUSED_L1=0" $OUT/r05b_ab.txt
bash tools/ab_projector_env.sh "EML_FUSED_L1=0 EML_SMALL_DA9=0" $OUT/r05b_ab.txt
cat $OUT/r05b_ab.txt
timeout 400 python tools/glue_audit.py joint 32 1 > $OUT/r05b_glue_joint.txt 2> $OUT/r05b_glue_joint.err
head -40 $OUT/r05b_glue_joint.txt
