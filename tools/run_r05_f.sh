#!/bin/bash
# round-5 GPU call F: same-box A/B of the whole round (the round-4 tree, build_exp/r04_tree, against this tree) on the
# regression / projector / joint legs (the round-4 tree is not in the history of this one -- it IS its history:
#   mkdir -p build_exp/r04_tree && git archive e9fe561 | tar -x -C build_exp/r04_tree && make -C build_exp/r04_tree/emlight_amd/csrc
# build_exp/ is git-ignored and travels to the GPU box with the snapshot), and SQ counters of the gather-GEMM with 16 loads, 10 loads and loads that cannot miss
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/r05f_round_ab.txt
for rep in 1 2; do
for tree in $REPO/build_exp/r04_tree $REPO; do
  timeout 600 python $tree/bench.py --steps 8 --warmup 3 --no_cpu_baseline --legs projector,joint 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s regression %7.2f img/s %8.3f ms | projector %7.2f img/s %8.3f ms | joint %7.2f img/s %8.3f ms (%.4f)' % ('$tree'.split('/')[-1], j['value'], j['ms_per_step'], j['projector']['value'], j['projector']['ms_per_step'], j['joint']['value'], j['joint']['ms_per_step'], j['joint']['roofline']['frac']))" >> $OUT/r05f_round_ab.txt
done
done
cat $OUT/r05f_round_ab.txt
rm -rf /tmp/pmcG
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcG -o g -- python $REPO/tools/bench_gather_share.py 32 probe > $OUT/r05f_probe.jsonl 2> $OUT/r05f_probe.err
python - > $OUT/r05f_gg2_pmc.txt <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for f in glob.glob("/tmp/pmcG/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gather_gemm2" not in r["Kernel_Name"]:
            continue
        out.setdefault(r["Dispatch_Id"], {"kernel": r["Kernel_Name"][:70]})[r["Counter_Name"]] = float(r["Counter_Value"])
# dispatches come in three groups of 23 (3 warm-up + 20 timed): 16 loads, 10 loads (row-shared), all taps -> pixel 0 (16, then 10)
ids = sorted(out, key=int)
groups = [ids[i:i + 23] for i in range(0, len(ids), 23)]
names = ["16 gathered loads per chunk", "10 (row-shared corners)", "16, every tap -> pixel 0 of the sample", "10, every tap -> pixel 0"]
for g, nm in zip(groups, names):
    acc = collections.defaultdict(float)
    for d in g[3:]:
        for k, v in out[d].items():
            if k != "kernel":
                acc[k] += v / len(g[3:])
    wc = acc["SQ_WAVE_CYCLES"]
    print("%-42s %s" % (nm, out[g[0]]["kernel"]))
    print("   wave-cycles %.4g: parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%; MFMA pipe busy %.1f %% of the "
          "SIMD-cycles; VMEM read instructions %.4g" % (wc, 100 * acc["SQ_WAIT_ANY"] / wc, 100 * acc["SQ_WAIT_INST_ANY"] / wc,
                                                         100 * acc["SQ_ACTIVE_INST_ANY"] / wc,
                                                         100 * acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (acc["GRBM_GUI_ACTIVE"] / 8 * 1024),
                                                         acc["SQ_INSTS_VMEM_RD"]))
PY
cat $OUT/r05f_probe.jsonl | cut -c1-250; cat $OUT/r05f_gg2_pmc.txt
