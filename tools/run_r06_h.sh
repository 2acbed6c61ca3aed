#!/bin/bash
# round 6, GPU call H: the D step's generator pass as a HIP graph -- its test, the trainer-level tests, then the joint / projector legs A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_projector.py tests/test_gpu_joint.py -m gpu -q -x -k "graph_equals or three_iterations or projector_iteration or joint" > gpurun_out/r06_h_tests.txt 2>&1; tail -6 gpurun_out/r06_h_tests.txt
for rep in 1 2; do
for v in 0 1; do
  EML_GRAPH_DSTEP=$v timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs projector,joint 2> gpurun_out/r06_h_bench_$v.err | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EML_GRAPH_DSTEP=$v  regression %.2f  projector %.2f img/s %.3f ms  joint %.2f img/s %.3f ms (%.4f)' % (j['value'], j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['step_frac_of_f32_mfma_peak']))" | tee -a gpurun_out/r06_h_ab_graph.txt
  grep -i "could not be captured\|Error" gpurun_out/r06_h_bench_$v.err | head -3
done
done
