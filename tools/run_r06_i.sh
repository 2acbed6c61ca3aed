#!/bin/bash
# round 6, GPU call I: the weight gradient with its dY tile by LDS-DMA (tests, A/B per layer and in the step); host syncs of a joint iteration
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_projector.py -m gpu -q -x -k "fused_kernels_vs_stock or natural_dispatch or lowres" > gpurun_out/r06_i_tests.txt 2>&1; tail -4 gpurun_out/r06_i_tests.txt
timeout 600 python tools/host_syncs.py > gpurun_out/r06_i_host_syncs.txt 2>&1; tail -40 gpurun_out/r06_i_host_syncs.txt
for rep in 1 2; do
for v in 1 0; do
  EML_WG_NODMA=$v timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --legs projector,joint 2> gpurun_out/r06_i_bench_$v.err | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
wg = [r for r in j['joint']['kernel_families'] if 'wgrad' in r['kernel'] or 'weight' in r['kernel']]
print('EML_WG_NODMA=$v  projector %.2f img/s %.3f ms  joint %.2f img/s %.3f ms (%.4f)' % (j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['step_frac_of_f32_mfma_peak']), [(r['kernel'][:40], r['ms_per_step'], r['tflops']) for r in wg][:3])" | tee -a gpurun_out/r06_i_ab_wgrad_dma.txt
done
done
