#!/bin/bash
# round 6, GPU call J: this tree against round 5's last commit (build_exp/r05_tree), same box, alternating
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
HERE=$(pwd)
: > gpurun_out/r06_j_round_ab.txt
for rep in 1 2; do
for t in r05 r06; do
  if [ $t = r05 ]; then D=$HERE/build_exp/r05_tree; else D=$HERE; fi
  ( cd $D && timeout 900 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs projector,joint 2> $HERE/gpurun_out/r06_j_$t.err ) | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t tree  regression %.2f img/s  projector %.2f img/s %.3f ms  joint %.2f img/s %.3f ms (%.4f)' % (j['value'], j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint'].get('step_frac_of_f32_mfma_peak') or j['joint']['roofline']['frac']))" | tee -a gpurun_out/r06_j_round_ab.txt
done
done
