#!/bin/bash
# round 6, GPU call A: same-box baseline of the round-5 tree (driver's command), the stream-capture bisect, the first RCCL runs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err
tail -c 1500 gpurun_out/r06_a_bench.json
timeout 900 python tools/capture_probe.py 8 64 > gpurun_out/r06_a_capture.txt 2>&1; tail -5 gpurun_out/r06_a_capture.txt
EML_DIST_SINGLE=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no_cpu_baseline --legs joint > gpurun_out/r06_a_rccl_single.json 2> gpurun_out/r06_a_rccl_single.err
echo "rccl single rc=$?"; tail -c 600 gpurun_out/r06_a_rccl_single.json; tail -5 gpurun_out/r06_a_rccl_single.err
EML_SHARE_GPUS=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no_cpu_baseline --legs none --batch 8 > gpurun_out/r06_a_rccl_two_on_one.json 2> gpurun_out/r06_a_rccl_two_on_one.err
echo "rccl two-on-one rc=$?"; tail -c 300 gpurun_out/r06_a_rccl_two_on_one.json; grep -i "error\|duplicate\|nccl" gpurun_out/r06_a_rccl_two_on_one.err | head -5
