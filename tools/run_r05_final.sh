#!/bin/bash
# round-5 closing GPU call: the whole -m gpu suite, the profile round (counters first), then the driver's bench command
# usage (from the repo, via gpurun): bash tools/run_r05_final.sh <commit>
COMMIT=${1:-unknown}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r05z_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r05z_pytest.txt
tail -3 $OUT/r05z_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05z_smoke.txt 2>&1; tail -1 $OUT/r05z_smoke.txt
SKIP_BENCH=1 bash tools/profile_round.sh r05 $COMMIT > $OUT/r05z_profile_round.log 2>&1
cd $REPO
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_bench.json 2> $OUT/r05_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
print("regression %.1f img/s %.2f ms frac %.3f | legs %s | traffic_source %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"],
      json.dumps(j.get("legs_summary")), j["roofline"].get("traffic_source")))
PY
