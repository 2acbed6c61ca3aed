#!/bin/bash
# usage: tools/pmc_kernels.sh <kernel-selector for bench_kernels.py>  (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp
SEL="$@"
rm -rf /tmp/pmcA /tmp/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmcA -o a -- python /root/repo/tools/bench_kernels.py $SEL > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcB -o b -- python /root/repo/tools/bench_kernels.py $SEL > /dev/null 2>&1
python /root/repo/tools/summarise_pmc.py /tmp/pmcA /tmp/pmcB
