#!/bin/bash
# round-5 GPU call E: the row-shared gather -- bit identity + per-layer A/B, spectral-norm launch times, projector tests, step A/B
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 600 python tools/bench_gather_share.py 32 > $OUT/r05e_gather_share.jsonl 2> $OUT/r05e_gather_share.err
cat $OUT/r05e_gather_share.jsonl | cut -c1-330
tail -3 $OUT/r05e_gather_share.err
timeout 900 python -m pytest tests/test_gpu_projector.py tests/test_gpu_joint.py tests/test_gpu_ddp_two_ranks.py -m gpu -q > $OUT/r05e_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05e_pytest.txt
tail -5 $OUT/r05e_pytest.txt
rm -f $OUT/r05e_ab.txt
bash tools/ab_joint_env.sh "EML_GG_NOSHARE=1" $OUT/r05e_ab.txt
bash tools/ab_projector_env.sh "EML_GG_NOSHARE=1" $OUT/r05e_ab.txt
cat $OUT/r05e_ab.txt
