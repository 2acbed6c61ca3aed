#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python tools/gpu_gaps.py joint 32 > gpurun_out/r06_l_gaps_joint.txt 2>&1; head -45 gpurun_out/r06_l_gaps_joint.txt
