#!/bin/bash
# round 6, GPU call D: parity of the footprint gather-GEMM, then what each piece of its loop costs (experiment builds)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_projector.py -m gpu -q -x -k "lowres" > gpurun_out/r06_d_tests.txt 2>&1; tail -12 gpurun_out/r06_d_tests.txt
timeout 900 python tools/exp/gg3_bench.py > gpurun_out/r06_d_gg3_variants.jsonl 2> gpurun_out/r06_d_gg3.err; cat gpurun_out/r06_d_gg3_variants.jsonl; tail -3 gpurun_out/r06_d_gg3.err
