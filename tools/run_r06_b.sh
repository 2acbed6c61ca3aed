#!/bin/bash
# round 6, GPU call B: the new tests on their own (timed), the capture bisect with the traceback, the one-rank RCCL dry run again
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_densenet.py -m gpu -q -x -s -k "cfg2_full_batch" ) > gpurun_out/r06_b_cfg2_full.txt 2>&1; tail -8 gpurun_out/r06_b_cfg2_full.txt
timeout 600 python -m pytest tests/test_gpu_projector.py -m gpu -q -x -k "more_pairs or fused_l1" > gpurun_out/r06_b_l1.txt 2>&1; tail -5 gpurun_out/r06_b_l1.txt
timeout 900 python tools/capture_probe.py 8 64 > gpurun_out/r06_b_capture.txt 2>&1; tail -40 gpurun_out/r06_b_capture.txt
EML_DIST_SINGLE=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no_cpu_baseline --legs joint > gpurun_out/r06_b_rccl_single.json 2> gpurun_out/r06_b_rccl_single.err
echo "rccl single rc=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_b_rccl_single.json").read().strip().splitlines()[-1])
print("regression", j["value"], j.get("rccl_ranks_seen"), j.get("collectives"))
print("joint", j["joint"]["value"], j["joint"]["ms_per_step"], j["joint"].get("collectives"))
for b in j["joint"]["other_breakdown"]:
    print(b)
print(j["config"]["runtime_env"])
PY
