#!/bin/bash
# round 6: the driver's round-end sequence on one box -- every GPU test, smoke, the bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -q -x ) > gpurun_out/r06_full_tests.txt 2>&1; tail -8 gpurun_out/r06_full_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_full_smoke.txt 2>&1; tail -3 gpurun_out/r06_full_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_full_bench.json 2> gpurun_out/r06_full_bench.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_full_bench.json").read().strip().splitlines()[-1])
print(json.dumps(j["legs_summary"]))
print({k: j["roofline"][k] for k in ("frac", "frac_operand_minimal", "operand_minimal_bytes_per_dispatch", "algorithmic_bytes_per_dispatch") if k in j["roofline"]})
print(j["config"]["runtime_env"], j.get("executed_gflop_per_image"), j.get("step_frac_executed"), j["joint"].get("step_frac_executed"))
PY
