#!/bin/bash
# round 5, late: library weight gradient on kept operands + the 2C = 512 SPADEs on the two-launch path (EML_WGRAD_LIB_KEPT,
# EML_SPADE512_TWO_LAUNCH): extend the GEMM record by the new shapes, then one steady-state joint iteration under rocprofv3
# with both knobs off / on (recorded GEMM selection in effect in both)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
EXTEND=1 bash tools/tune_gemms.sh 30 30
cp $OUT/tunableop_results0.csv emlight_amd/tuned_gemms_gfx950.csv
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/kt$v
  EML_WGRAD_LIB_KEPT=$v EML_SPADE512_TWO_LAUNCH=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$v -o k -- python $REPO/bench.py --workload joint --steps 2 --warmup 1 > $OUT/r05v_kt$v.log 2>&1
  python $REPO/tools/steady_step.py $(find /tmp/kt$v -name '*kernel_trace.csv' | head -1) conv0_fwd_kernel 1 2 > $OUT/r05v_steady_$v.csv
done
python - > $OUT/r05v_ab.txt <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for v in (0, 1):
    rows = list(csv.reader(open("%s/r05v_steady_%d.csv" % (out, v))))
    body = rows[2:]
    def tot(pred):
        sel = [r for r in body if pred(r[0])]
        return sum(float(r[2]) for r in sel) / 1e3, sum(int(r[1]) for r in sel)
    lib = tot(lambda k: k.startswith("Cijk") or "rocblas" in k.lower())
    wg = tot(lambda k: "sphere_conv_wgrad_fused" in k)
    gg = tot(lambda k: "gg2::" in k)
    ic = tot(lambda k: "im2col" in k or "col2im" in k)
    sp = tot(lambda k: "spade_norm_modulate" in k)
    print("knobs=%d: all kernels %.2f ms | library GEMM %.2f ms / %d | fused wgrad %.2f ms / %d | gg2 %.2f ms / %d | im2col+col2im %.2f ms / %d | spade modulate %.2f ms / %d"
          % ((v, sum(float(r[2]) for r in body) / 1e3) + lib + wg + gg + ic + sp))
PY
cat $OUT/r05v_ab.txt
