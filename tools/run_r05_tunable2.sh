#!/bin/bash
# TunableOp A/B by kernel time: one steady-state joint iteration under rocprofv3 with the default library selection and with
# the recorded one (build_exp/tunableop_results0.csv)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTORCH_TUNABLEOP_FILENAME=$REPO/build_exp/tunableop_results.csv PYTORCH_TUNABLEOP_TUNING=0
for v in 0 1; do
  rm -rf /tmp/kt$v
  PYTORCH_TUNABLEOP_ENABLED=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$v -o k -- python $REPO/bench.py --workload joint --steps 2 --warmup 1 > $OUT/r05t_kt$v.log 2>&1
  python $REPO/tools/steady_step.py $(find /tmp/kt$v -name '*kernel_trace.csv' | head -1) conv0_fwd_kernel 1 2 > $OUT/r05t_steady_$v.csv
  head -2 $OUT/r05t_steady_$v.csv | tail -1 | cut -c1-160
done
python - <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
def load(v):
    rows = list(csv.reader(open("%s/r05t_steady_%d.csv" % (out, v))))[2:]
    lib = sum(float(r[2]) for r in rows if r[0].startswith("Cijk") or "rocblas" in r[0].lower())
    n = sum(int(r[1]) for r in rows if r[0].startswith("Cijk") or "rocblas" in r[0].lower())
    tot = sum(float(r[2]) for r in rows)
    return lib, n, tot
for v in (0, 1):
    print("TUNABLEOP=%d: library GEMM kernels %.1f us in %d launches; all kernels %.1f us" % ((v,) + load(v)))
PY
