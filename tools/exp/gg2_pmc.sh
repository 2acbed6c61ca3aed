#!/bin/bash
# SQ / clock counters of the gather-GEMM kernels on ONE layer shape (runs on the GPU box):
#   tools/exp/gg2_pmc.sh "<substring of the layer name + role>" <out.txt>
SHAPE=${1:-"up_3 g|b 128->128 @128x256 fwd"}
OUTF=${2:-gpurun_out/gg2_pmc.txt}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC
export GG2_ONLY="$SHAPE"
CMD="python $REPO/tools/exp/gg2_bench.py 5"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcA -o a -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcB -o b -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES --output-format csv -d /tmp/pmcC -o c -- $CMD > /dev/null 2>&1
python - > $REPO/$OUTF <<PY
import csv, glob, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def key(r):
    n = r["Kernel_Name"]
    if "sphere_conv" not in n and "gather_gemm2" not in n:
        return None
    m = re.search(r"((sphere_conv_\w+_kernel|gather_gemm2_kernel)<[^>]*>)", n)
    return (m.group(1) if m else n[:60]) + " lds=" + r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")) + " vgpr=" + r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?"))
for d in ("/tmp/pmcA", "/tmp/pmcB", "/tmp/pmcC"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k:
                dur[k.split(" lds=")[0]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("shape:", "$SHAPE")
for k, cs in out.items():
    d = sorted(dur.get(k.split(" lds=")[0], [0]))
    print(k, " median_ns=%.0f" % d[len(d) // 2])
    for c, v in sorted(cs.items()):
        print("   %-28s %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
head -150 $REPO/$OUTF
