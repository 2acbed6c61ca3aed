#!/bin/bash
# GPU side: block 1's geometry under every build_exp/lib_tp_*.so (tools/exp/c3tp_variants.sh) and the product library
REPO=${GRAFT_REPO_ROOT:-/root/repo}
echo "== product"; C3TP_QUICK=1 python $REPO/tools/bench_c3tp.py ${1:-64} 2>&1 | grep -v amdgpu.ids
for lib in $REPO/build_exp/lib_tp_*.so; do
  echo "== $(basename $lib)"; EML_LIB_PATH=$lib C3TP_QUICK=1 python $REPO/tools/bench_c3tp.py ${1:-64} 2>&1 | grep -v amdgpu.ids
done
