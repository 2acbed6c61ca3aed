#!/bin/bash
# SQ counters of the footprint gather-GEMM's experiment builds on ONE forward shape (runs on the GPU box):
#   tools/exp/gg3_pmc.sh "<substring of the layer name>" <out.txt> variant...
SHAPE=${1:-"up_0 1024->512"}
OUTF=${2:-gpurun_out/gg3_pmc.txt}
shift; shift
VARIANTS=${@:-base bare nocorner notab nofp}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export GG3_ONLY="$SHAPE"
: > $REPO/$OUTF
for V in $VARIANTS; do
  export GG3_VARIANT=$V
  rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC
  CMD="python $REPO/tools/exp/gg3_bench.py 5"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcA -o a -- $CMD > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcB -o b -- $CMD > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES --output-format csv -d /tmp/pmcC -o c -- $CMD > /dev/null 2>&1
  python - "$V" >> $REPO/$OUTF <<'PY'
import csv, glob, collections, sys
out = collections.defaultdict(list)
dur = []
for d in ("/tmp/pmcA", "/tmp/pmcB", "/tmp/pmcC"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gather_gemm3" in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gather_gemm3" in r["Kernel_Name"]:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
dur.sort()
m = {c: sum(v) / len(v) for c, v in out.items()}
print("variant %-10s median_ns=%.0f  dispatches=%d" % (sys.argv[1], dur[len(dur) // 2] if dur else 0, len(dur)))
wc = m.get("SQ_WAVE_CYCLES", 0) or 1
print("   wave-cycles %.4g: parked %.1f %%, issue-stalled %.1f %% (of which LDS-issue %.1f %%), issuing %.1f %%; MFMA busy cycles %.4g; busy cycles %.4g"
      % (wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc,
         100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("SQ_BUSY_CYCLES", 0)))
for c in sorted(m):
    print("   %-28s %.5g" % (c, m[c]))
PY
done
cat $REPO/$OUTF
