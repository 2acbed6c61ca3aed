#!/bin/bash
# Round evidence, one gpurun call:  tools/profile_round.sh <tag> <commit>   (e.g. r02 $(git rev-parse --short HEAD);
# the GPU box has no .git, so the commit the numbers belong to is passed in by the caller)
# Writes gpurun_out/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_summary.csv (+ .meta.json),
# <tag>_projector_kernel_stats.csv, <tag>_joint_kernel_stats.csv, <tag>_{bench,projector,joint}_steady_step.csv.  SKIP_BENCH=1 / SKIP_PROJ=1 skip parts.
TAG=${1:-r02}
COMMIT=${2:-unknown}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# counters FIRST, and their summary into profiles/ of this copy of the repo: the bench line's roofline.traffic is read from the
# newest profiles/rNN_pmc_summary.csv, so the line below is stamped with THIS commit's collection (hygiene)
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc_$(echo $C | cut -d' ' -f1)
  rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o p -- python $REPO/bench.py --steps 1 --warmup 1 --no_cpu_baseline --legs families > $OUT/${TAG}_pmc.log 2>&1
done
python $REPO/tools/summarise_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $OUT/${TAG}_pmc_summary.csv
echo "{\"collected_at_commit\": \"$COMMIT\", \"command\": \"rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE} -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --legs families (three separate passes), tools/summarise_pmc.py\"}" > $OUT/${TAG}_pmc_summary.meta.json
cp $OUT/${TAG}_pmc_summary.csv $OUT/${TAG}_pmc_summary.meta.json $REPO/profiles/
head -40 $OUT/${TAG}_pmc_summary.csv
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python $REPO/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
fi
rm -rf /tmp/kst
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $REPO/bench.py --steps 3 --warmup 1 --no_cpu_baseline --legs families,sinkhorn,rasteriser > $OUT/${TAG}_kst.log 2>&1
cp $(find /tmp/kst -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# one steady-state iteration (the 3rd of the trace) from the per-dispatch trace: the table to read per-step times from
python $REPO/tools/steady_step.py $(find /tmp/kst -name '*kernel_trace.csv' | head -1) conv0_fwd_ 1 2 > $OUT/${TAG}_bench_steady_step.csv
if [ -z "$SKIP_PROJ" ]; then
  for WL in projector joint; do
    rm -rf /tmp/kst_$WL
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$WL -o k -- python $REPO/bench.py --workload $WL --steps 2 --warmup 1 > $OUT/${TAG}_${WL}_kst.log 2>&1
    cp $(find /tmp/kst_$WL -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${WL}_kernel_stats.csv
    # the 3rd iteration of the trace = the 2nd timed step of the variant WITH the VGG term (markers: the encoder's first kernel /
    # the fold of the fused L1 terms, which a projector step launches twice: feature matching, VGG)
    if [ $WL = joint ]; then MK="conv0_fwd_ 1"; else MK="l1_pairs_fold_kernel 2"; fi
    python $REPO/tools/steady_step.py $(find /tmp/kst_$WL -name '*kernel_trace.csv' | head -1) $MK 2 > $OUT/${TAG}_${WL}_steady_step.csv
  done
fi
