#!/bin/bash
# round-5 GPU call C: the whole -m gpu suite (no -x: every failure listed), then a kernel trace of the joint step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q > $OUT/r05c_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05c_pytest.txt
tail -6 $OUT/r05c_pytest.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst_joint
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_joint -o k -- python $REPO/bench.py --workload joint --steps 2 --warmup 1 > $OUT/r05c_joint_kst.log 2>&1
cp $(find /tmp/kst_joint -name '*kernel_stats.csv' | head -1) $OUT/r05c_joint_kernel_stats.csv
tail -2 $OUT/r05c_joint_kst.log | cut -c1-300
