#!/bin/bash
# Rebuild ONLY vn_fused16.o with extra compiler flags and link it with the other objects of the working tree into
# varnet_amd/libvarnet_hip_<name>.so (flag experiments on the hot kernel: tools/ab_perf.py a,b,c).
#   tools/build_fused16_variant.sh <name> "<extra flags>"
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/varnet_amd/csrc
make -j8 > /dev/null
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $extra -c vn_fused16.hip -o $tmp/vn_fused16.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 vn_api.o vn_generic.o vn_pointwise.o vn_fused.o $tmp/vn_fused16.o vn_dedup.o vn_layered.o vn_wide.o vn_gemm.o -o ../libvarnet_hip_$name.so
rm -rf $tmp
echo built libvarnet_hip_$name.so
