#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_dense_kernels.py tests/test_gpu_densenet.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r06_top_tests.txt
cat $OUT/r06_top_tests.txt
bash tools/ab_top.sh 2>&1 | tee $OUT/r06_ab_top.txt
