#!/bin/bash
# round-5 GPU call A: parity of the new spectral-norm kernels (whole -m gpu suite), glue audit, per-shape op profile, joint leg
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r05a_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05a_pytest.txt
tail -5 $OUT/r05a_pytest.txt
timeout 400 python tools/glue_audit.py joint 32 1 > $OUT/r05a_glue_joint.txt 2> $OUT/r05a_glue_joint.err
timeout 400 python tools/projector_op_profile.py 32 > $OUT/r05a_projector_ops.txt 2> $OUT/r05a_projector_ops.err
timeout 600 python bench.py --legs joint --steps 5 --warmup 2 --no_cpu_baseline > $OUT/r05a_bench_joint.json 2> $OUT/r05a_bench_joint.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a_bench_joint.json').read().strip().splitlines()[-1])
print('regression', d['value'], d['ms_per_step'])
j=d.get('joint',{})
print('joint', j.get('value'), j.get('ms_per_step'), j.get('roofline',{}).get('frac'), j.get('without_vgg'))
for o in j.get('other_breakdown',[]): print(o)
PY
