#!/bin/bash
# usage: tools/exp_build.sh <name> <extra hipcc flags...>  -> build_exp/lib_<name>.so (experiment builds; use with EML_LIB_PATH)
set -e
NAME=$1; shift
mkdir -p /root/repo/build_exp/$NAME
cd /root/repo/emlight_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o /root/repo/build_exp/$NAME/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /root/repo/build_exp/$NAME/*.o -o /root/repo/build_exp/lib_$NAME.so
rm -rf /root/repo/build_exp/$NAME
