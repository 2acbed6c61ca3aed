#!/bin/bash
# round-5 GPU call G: fused spectral norm on the crop encoder's convolutions (tests), split-K knob of the fused wgrad (A/B)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_projector.py tests/test_gpu_joint.py tests/test_projector_golden.py -m gpu -q > $OUT/r05g_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05g_pytest.txt
tail -5 $OUT/r05g_pytest.txt
rm -f $OUT/r05g_ab.txt
bash tools/ab_joint_env.sh "EML_WGRAD_WGS=1024" $OUT/r05g_ab.txt
bash tools/ab_joint_env.sh "EML_WGRAD_WGS=4096" $OUT/r05g_ab.txt
cat $OUT/r05g_ab.txt
timeout 400 python tools/glue_audit.py joint 32 1 > $OUT/r05g_glue_joint.txt 2> $OUT/r05g_glue_joint.err
head -36 $OUT/r05g_glue_joint.txt
