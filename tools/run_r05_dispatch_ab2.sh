#!/bin/bash
# round 5, late: EML_WGRAD_LIB_KEPT and EML_WGRAD_DIRECT one at a time (kernel time of one steady-state joint iteration,
# recorded GEMM selection in effect, the record extended by the new orientation first)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
EXTEND=1 bash tools/tune_gemms.sh 30 30
cp $OUT/tunableop_results0.csv emlight_amd/tuned_gemms_gfx950.csv
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/r05w_ab.txt
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  rm -rf /tmp/kt
  EML_WGRAD_LIB_KEPT=$1 EML_WGRAD_DIRECT=$2 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $REPO/bench.py --workload joint --steps 2 --warmup 1 > $OUT/r05w_kt.log 2>&1
  python $REPO/tools/steady_step.py $(find /tmp/kt -name '*kernel_trace.csv' | head -1) conv0_fwd_kernel 1 2 > $OUT/r05w_steady_$1$2.csv
  python - $OUT/r05w_steady_$1$2.csv "EML_WGRAD_LIB_KEPT=$1 EML_WGRAD_DIRECT=$2" >> $OUT/r05w_ab.txt <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
body = rows[2:]
def tot(pred):
    sel = [r for r in body if pred(r[0])]
    return sum(float(r[2]) for r in sel) / 1e3, sum(int(r[1]) for r in sel)
print("%s: all kernels %.2f ms / %d | library GEMM %.2f ms / %d | fused wgrad %.2f ms / %d | ATen copies %.2f ms / %d"
      % ((sys.argv[2], sum(float(r[2]) for r in body) / 1e3, sum(int(r[1]) for r in body)) + tot(lambda k: k.startswith("Cijk") or "rocblas" in k.lower())
         + tot(lambda k: "sphere_conv_wgrad_fused" in k) + tot(lambda k: "direct_copy" in k)))
PY
done
cat $OUT/r05w_ab.txt
