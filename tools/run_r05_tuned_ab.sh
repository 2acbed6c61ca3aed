#!/bin/bash
# the recorded library-GEMM selection (emlight_amd/tuned_gemms_gfx950.csv) against the library defaults, same box:
# kernel time of one steady-state joint iteration (rocprofv3), then the bench legs, EML_TUNED_GEMMS=0 / 1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "
import sys; sys.path.insert(0, '$REPO')
from emlight_amd import _lib, _gemm_selection
_lib.lib(); print(_gemm_selection.status())" > $OUT/r05u_ab.txt 2>&1
for v in 0 1; do
  rm -rf /tmp/kt$v
  EML_TUNED_GEMMS=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$v -o k -- python $REPO/bench.py --workload joint --steps 2 --warmup 1 > $OUT/r05u_kt$v.log 2>&1
  python $REPO/tools/steady_step.py $(find /tmp/kt$v -name '*kernel_trace.csv' | head -1) conv0_fwd_kernel 1 2 > $OUT/r05u_steady_$v.csv
done
python - >> $OUT/r05u_ab.txt <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for v in (0, 1):
    rows = list(csv.reader(open("%s/r05u_steady_%d.csv" % (out, v))))
    body = rows[2:]
    sel = [r for r in body if r[0].startswith("Cijk") or "rocblas" in r[0].lower()]
    print("EML_TUNED_GEMMS=%d: library GEMM kernels %.1f us in %d launches; all kernels %.1f us; %s" % (
        v, sum(float(r[2]) for r in sel), sum(int(r[1]) for r in sel), sum(float(r[2]) for r in body), rows[1][0]))
PY
cd $REPO
for v in 0 1 0 1; do
  EML_TUNED_GEMMS=$v timeout 600 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs projector,joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EML_TUNED_GEMMS=$v  projector %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % (j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['roofline']['frac']))" >> $OUT/r05u_ab.txt
done
cat $OUT/r05u_ab.txt
