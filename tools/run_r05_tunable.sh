#!/bin/bash
# round-5 experiment: PyTorch TunableOp on the library GEMMs of the projector / joint step (the wide low-resolution layers'
# im2col products, 59 ms of a joint step at 130-135 TF/s): tune once (bounded), then A/B with the recorded selection
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
export PYTORCH_TUNABLEOP_FILENAME=$OUT/tunableop_results.csv
rm -f $OUT/tunableop_results*.csv
T0=$(date +%s)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${TUNE_MS:-15} PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=${TUNE_IT:-10} PYTORCH_TUNABLEOP_VERBOSE=0 \
  timeout 1200 python bench.py --workload joint --steps 1 --warmup 1 > $OUT/r05t_tune.log 2>&1
echo "tuning run: rc=$? $(( $(date +%s) - T0 )) s; $(cat $OUT/tunableop_results*.csv 2>/dev/null | wc -l) lines" | tee $OUT/r05t_ab.txt
for v in 0 1 0 1; do
  PYTORCH_TUNABLEOP_ENABLED=$v PYTORCH_TUNABLEOP_TUNING=0 timeout 600 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs projector,joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TUNABLEOP=$v  projector %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % (j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['roofline']['frac']))" >> $OUT/r05t_ab.txt
done
cat $OUT/r05t_ab.txt
