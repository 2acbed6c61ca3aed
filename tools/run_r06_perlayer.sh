#!/bin/bash
# Per-dispatch durations and HBM counters of the encoder's 1x1 kernels over one regression step (tools/per_layer_trace.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline --legs none > $OUT/r06_perlayer_kt.log 2>&1
python $REPO/tools/per_layer_trace.py $(find /tmp/kt -name '*kernel_trace.csv' | head -1) conv0_fwd_ 2 conv1x1_ conv3x3_ > $OUT/r06_perlayer_durations.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pc_$C -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no_cpu_baseline --legs none > $OUT/r06_perlayer_pmc.log 2>&1
  python $REPO/tools/per_layer_trace.py $(find /tmp/pc_$C -name '*counter_collection.csv' | head -1) conv0_fwd_ 1 conv1x1_ > $OUT/r06_perlayer_$C.txt
done
tail -5 $OUT/r06_perlayer_durations.txt
