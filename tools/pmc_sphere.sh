#!/bin/bash
# SQ wait / busy counters of the fused SphereConv kernels (runs on the GPU box):  tools/pmc_sphere.sh [C O H W B]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcA -o a -- python $REPO/tools/bench_sphere_fused.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcB -o b -- python $REPO/tools/bench_sphere_fused.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_WAVES --output-format csv -d /tmp/pmcC -o c -- python $REPO/tools/bench_sphere_fused.py "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pmcA", "/tmp/pmcB", "/tmp/pmcC"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sphere_conv" not in r["Kernel_Name"]:
                continue
            m = re.search(r"(sphere_conv_\w+_kernel<[^>]*>)", r["Kernel_Name"])
            out[m.group(1) if m else r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
