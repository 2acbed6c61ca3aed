#!/bin/bash
# experiment builds of csrc/sphere_conv_lowres.hip (gather_gemm3.h) with one piece of its loop removed each -- wrong results,
# timed by tools/exp/gg3_bench.py to see what each piece costs.  Runs here (cross-compiles); the .so files travel with gpurun.
set -e
cd "$(dirname "$0")/../.." && mkdir -p build_exp
for v in base:-DGG3_BASE nobdma:-DGG3_NOBDMA notab:-DGG3_NOTAB nocorner:-DGG3_NOCORNER nofp:-DGG3_NOFP nobar:-DGG3_NOBAR \
         bare:"-DGG3_NOBDMA -DGG3_NOTAB -DGG3_NOCORNER -DGG3_NOFP" scalar:-DGG3_SCALARCOMBINE bunched:-DGG3_BUNCHED scalarbunched:"-DGG3_SCALARCOMBINE -DGG3_BUNCHED" $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags emlight_amd/csrc/sphere_conv_lowres.hip -o build_exp/libgg3_$name.so &
done
wait
ls -la build_exp/libgg3_*.so
