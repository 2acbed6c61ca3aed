#!/bin/bash
# round 6, GPU call C: the footprint gather-GEMM -- parity tests, then per-layer timing against round 5's dispatch; the capture probe again
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_projector.py -m gpu -q -x -k "lowres or more_pairs" > gpurun_out/r06_c_tests.txt 2>&1; tail -15 gpurun_out/r06_c_tests.txt
timeout 900 python tools/bench_lowres.py > gpurun_out/r06_c_lowres.jsonl 2> gpurun_out/r06_c_lowres.err; cat gpurun_out/r06_c_lowres.jsonl; tail -3 gpurun_out/r06_c_lowres.err
timeout 600 python tools/capture_probe.py 8 64 > gpurun_out/r06_c_capture.txt 2>&1; grep -v "^      \|^    File" gpurun_out/r06_c_capture.txt | tail -12
