#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_projector.py -m gpu -q -k "lowres" > gpurun_out/r06_m_tests.txt 2>&1; tail -3 gpurun_out/r06_m_tests.txt
timeout 900 python tools/exp/gg3_bench.py > gpurun_out/r06_m_gg3.jsonl 2> gpurun_out/r06_m_gg3.err; python - <<'PY'
import json
for l in open("gpurun_out/r06_m_gg3.jsonl"):
    j = json.loads(l)
    if j.get("role") == "backward":
        print(j["layer"], "bwd:", {k: (v["ms"], v["tflops"]) for k, v in j.items() if isinstance(v, dict) and k.startswith("dgrad")})
    else:
        print(j["layer"], "split", j["split"], {k: v.get("tflops", v.get("total_tflops")) for k, v in j.items() if isinstance(v, dict)}, "lib gemm", j["library"]["gemm_tflops"])
PY
tail -2 gpurun_out/r06_m_gg3.err
