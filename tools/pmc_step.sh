#!/bin/bash
# SQ issue / wait counters of the regression step's kernels (one gpurun call): tools/pmc_step.sh <tag>
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqA /tmp/sqB
CMD="python $REPO/bench.py --steps 1 --warmup 1 --no_cpu_baseline --legs none"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/sqA -o a -- $CMD > $OUT/${TAG}_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/sqB -o b -- $CMD >> $OUT/${TAG}_sq.log 2>&1
python $REPO/tools/summarise_pmc.py /tmp/sqA /tmp/sqB > $OUT/${TAG}_sq_summary.csv
grep -E "^kernel|conv1x1|conv3x3|transition" $OUT/${TAG}_sq_summary.csv
