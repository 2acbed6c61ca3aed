#!/bin/bash
# round-5 GPU call H: is the fused spectral norm on the crop encoder's nn.Conv2d layers a win? (A/B), other_breakdown of the joint leg
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
rm -f $OUT/r05h_ab.txt
bash tools/ab_joint_env.sh "EML_SN_CONV2D=0" $OUT/r05h_ab.txt
cat $OUT/r05h_ab.txt
timeout 600 python bench.py --legs joint --steps 5 --warmup 2 --no_cpu_baseline > $OUT/r05h_bench_joint.json 2> $OUT/r05h_bench_joint.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05h_bench_joint.json').read().strip().splitlines()[-1])
print('regression', d['value'], d['ms_per_step'])
j=d.get('joint',{})
print('joint', j.get('value'), j.get('ms_per_step'), j.get('roofline',{}).get('frac'), j.get('without_vgg'))
for o in j.get('other_breakdown',[]): print(o)
for o in j.get('kernel_families',[])[:12]: print(o['ms_per_step'], o['tflops'], o['kernel'][:80])
PY
