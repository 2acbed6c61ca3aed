#!/bin/bash
# round 5, last GPU call: the discriminator step's generator pass as a replayed HIP graph (EML_GRAPH_DSTEP).  Stage A: its test and
# an alternating A/B of the joint leg; the switch is kept ON only if the test passes, the pass really ran as a graph and the
# step got >= 0.7 ms faster.  Stage B: the closing evidence (tools/run_r05_final.sh) with that decision exported.
COMMIT=${1:-unknown}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_projector.py -m gpu -q -x -k "graph_equals_eager" > $OUT/r05g_pytest.txt 2>&1
TEST_RC=$?
tail -3 $OUT/r05g_pytest.txt
rm -f $OUT/r05g_ab.txt
for v in 0 1 0 1; do
  EML_GRAPH_DSTEP=$v timeout 300 python bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs joint 2>$OUT/r05g_bench_$v.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EML_GRAPH_DSTEP=$v  joint %7.2f img/s %8.3f ms (%.4f)  pass=%s' % (j['joint']['value'], j['joint']['ms_per_step'], j['joint']['roofline']['frac'], j['joint']['config'].get('dstep_generator_pass')))" >> $OUT/r05g_ab.txt
done
cat $OUT/r05g_ab.txt
DECISION=$(python - $TEST_RC <<'PY'
import re, sys, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r05g_ab.txt"
ms = {0: [], 1: []}
graph = True
try:
    for l in open(out):
        m = re.match(r"EML_GRAPH_DSTEP=(\d)\s+joint\s+[\d.]+ img/s\s+([\d.]+) ms .*pass=(\w+)", l)
        if m:
            ms[int(m.group(1))].append(float(m.group(2)))
            if m.group(1) == "1" and m.group(3) != "graph":
                graph = False
    ok = sys.argv[1] == "0" and graph and len(ms[0]) == 2 and len(ms[1]) == 2 and (sum(ms[0]) - sum(ms[1])) / 2 >= 0.7
except Exception:
    ok = False
print(1 if ok else 0)
PY
)
echo "decision: EML_GRAPH_DSTEP=$DECISION (test rc=$TEST_RC)" | tee -a $OUT/r05g_ab.txt
EML_GRAPH_DSTEP=$DECISION bash tools/run_r05_final.sh $COMMIT
