#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 600 python tools/host_syncs.py > gpurun_out/r06_k_host_syncs.txt 2>&1; tail -45 gpurun_out/r06_k_host_syncs.txt
