#!/bin/bash
# HIP_FORCE_DEV_KERNARG=0 / 1 (kernel arguments in device memory): bench legs alternating, then one traced joint iteration each
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
rm -f $OUT/r05k_ab.txt
for v in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 600 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HIP_FORCE_DEV_KERNARG=$v  regression %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % (j['value'], j['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['roofline']['frac']))" >> $OUT/r05k_ab.txt
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/kt
  HIP_FORCE_DEV_KERNARG=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --workload joint --steps 2 --warmup 1 > $OUT/r05k_kt.log 2>&1
  python $REPO/tools/steady_step.py $(find /tmp/kt -name '*kernel_trace.csv' | head -1) conv0_fwd_kernel 1 2 | head -2 | tail -1 | cut -c1-140 | sed "s/^/HIP_FORCE_DEV_KERNARG=$v /" >> $OUT/r05k_ab.txt
done
cat $OUT/r05k_ab.txt
