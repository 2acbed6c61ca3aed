#!/bin/bash
# Actual HBM traffic per kernel of the joint step (tools/pmc_traffic.py): which kernels are at the streaming ceiling, which are not
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload joint --steps 1 --warmup 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tj_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/tj_$C -o p -- $CMD > $OUT/r06_traffic_$C.log 2>&1
done
rm -rf /tmp/tj_kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tj_kt -o k -- $CMD > $OUT/r06_traffic_kt.log 2>&1
python $REPO/tools/pmc_traffic.py /tmp/tj_FETCH_SIZE /tmp/tj_WRITE_SIZE /tmp/tj_kt 0.2 > $OUT/r06_joint_traffic.txt
head -60 $OUT/r06_joint_traffic.txt
