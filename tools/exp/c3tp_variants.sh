#!/bin/bash
# Experiment builds of csrc/dense_fwd_tp.hip (run HERE, no GPU needed): tools/exp/c3tp_variants.sh name "flags" [name "flags" ...]
# -> build_exp/lib_tp_<name>.so (the other objects are the product's); measured by tools/bench_c3tp.py under EML_LIB_PATH.
set -e
cd /root/repo/emlight_amd/csrc
make -s > /dev/null
mkdir -p /root/repo/build_exp
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c dense_fwd_tp.hip -o /tmp/tp_$NAME.o
  OBJS=$(ls *.o | grep -v dense_fwd_tp.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/tp_$NAME.o -o /root/repo/build_exp/lib_tp_$NAME.so
  echo built lib_tp_$NAME.so
done
