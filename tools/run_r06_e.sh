#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_projector.py -m gpu -q -k "lowres" > gpurun_out/r06_e_tests.txt 2>&1; tail -6 gpurun_out/r06_e_tests.txt
GG3_VARIANTS_OFF=1 timeout 900 python tools/exp/gg3_bench.py > gpurun_out/r06_e_gg3.jsonl 2> gpurun_out/r06_e_gg3.err; grep backward gpurun_out/r06_e_gg3.jsonl; tail -3 gpurun_out/r06_e_gg3.err
