#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
bash tools/exp/gg3_pmc.sh "up_0 1024->512" gpurun_out/r06_f_gg3_pmc.txt base bare nocorner notab nofp > /dev/null 2>&1
grep -A1 "^variant" gpurun_out/r06_f_gg3_pmc.txt
