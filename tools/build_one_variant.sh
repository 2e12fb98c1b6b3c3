#!/bin/bash
# Rebuild ONE translation unit of varnet_amd/csrc with extra compiler flags and link it with the other objects of the working tree
# into varnet_amd/libvarnet_hip_<name>.so (flag / variant experiments on one kernel; VARNET_HIP_LIB=... or tools/ab_perf.py).
#   tools/build_one_variant.sh vn_split16 <name> "<extra flags>"
set -e
unit=$1; name=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/varnet_amd/csrc
make -j8 > /dev/null
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -I. $extra -c $unit.hip -o $tmp/$unit.o
objs=$(ls *.o | grep -v "^$unit.o$\|^vn_api_x.o$\|^vn_fused.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $tmp/$unit.o -o ../libvarnet_hip_$name.so
rm -rf $tmp
echo built libvarnet_hip_$name.so
