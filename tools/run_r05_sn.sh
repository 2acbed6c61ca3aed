#!/bin/bash
# round-5 GPU call: the batched spectral norm (spectral_precompute) -- its tests, then the projector / joint legs with
# EML_SN_BATCH=0 / 1 on the same box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_projector.py -m gpu -q -x -k "spectral or generator_step or joint" > $OUT/r05s_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05s_pytest.txt
tail -5 $OUT/r05s_pytest.txt
rm -f $OUT/r05s_ab.txt
for v in 0 1 0 1; do
  EML_SN_BATCH=$v timeout 600 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs projector,joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EML_SN_BATCH=$v  projector %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % (j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['roofline']['frac']))" >> $OUT/r05s_ab.txt
done
cat $OUT/r05s_ab.txt
