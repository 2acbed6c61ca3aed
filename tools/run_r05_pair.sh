#!/bin/bash
# round-5 GPU call: the pair pass of the encoder's 1x1 backward (EML_PAIR_PASS=1) -- the encoder's end-to-end tests with it on
# (FAST=1: only the golden train step), then the regression step with it off / on (same box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
SEL=""; [ -n "$FAST" ] && SEL="-k golden"
EML_PAIR_PASS=1 timeout 900 python -m pytest tests/test_gpu_densenet.py -m gpu -q -x $SEL > $OUT/r05p_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05p_pytest.txt
tail -3 $OUT/r05p_pytest.txt
rm -f $OUT/r05p_ab.txt
for v in 0 1 1; do
  EML_PAIR_PASS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs none 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EML_PAIR_PASS=$v  %7.2f img/s %8.3f ms' % (j['value'], j['ms_per_step']))" >> $OUT/r05p_ab.txt
done
cat $OUT/r05p_ab.txt
