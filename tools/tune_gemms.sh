#!/bin/bash
# Record the fastest rocBLAS / hipBLASLt solution for every plain GEMM shape of the projector and joint steps (BASELINE
# shapes, 32 per GPU) with PyTorch's TunableOp: gpurun_out/tunableop_results0.csv, to be committed as
# emlight_amd/tuned_gemms_gfx950.csv (emlight_amd/_gemm_selection.py).   bash tools/tune_gemms.sh [ms per solution] [iterations]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
rm -f $OUT/tunableop_results*.csv
# EXTEND=1: start from the committed record and time only the shapes it lacks (a dispatch change added a few products)
[ -n "$EXTEND" ] && cp emlight_amd/tuned_gemms_gfx950.csv $OUT/tunableop_results0.csv
T0=$(date +%s)
PYTORCH_TUNABLEOP_FILENAME=$OUT/tunableop_results.csv PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${1:-30} PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=${2:-30} PYTORCH_TUNABLEOP_VERBOSE=0 \
  timeout 1500 python bench.py --steps 1 --warmup 1 --no_cpu_baseline --legs projector,joint > $OUT/tune_gemms.log 2>&1
echo "tuning: rc=$? $(( $(date +%s) - T0 )) s; $(cat $OUT/tunableop_results*.csv 2>/dev/null | wc -l) lines"
