#!/bin/bash
# Rebuild ONLY vn_fused16.o with extra compiler flags and link it with the other objects of the working tree into
# varnet_amd/libvarnet_hip_<name>.so (flag experiments on the hot kernel: tools/ab_perf.py a,b,c).
#   tools/build_fused16_variant.sh <name> "<extra flags>"
# The diagnostic blocks of the hot kernel (-DVN_STAMPS=1/2/3 phase stamps, -DVN_FIXSTAMPS, the -DVN_ABL_* ablations) are NOT in
# the product source any more (round 6): they live in tools/diag/vn_fused16_diag.patch, which this script applies to a COPY of
# vn_fused16.hip whenever the flags name one of them (tools/diag/strip_diag.py is how the product source was derived from the
# instrumented one; the two preprocess to the same device code when no diagnostic macro is defined).
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/varnet_amd/csrc
make -j8 > /dev/null
tmp=$(mktemp -d)
src=vn_fused16.hip
if echo "$extra" | grep -q "VN_STAMPS\|VN_FIXSTAMPS\|VN_ABL_"; then
  mkdir -p $tmp/varnet_amd/csrc && cp vn_fused16.hip $tmp/varnet_amd/csrc/
  (cd $tmp && patch -s -p1 < $root/tools/diag/vn_fused16_diag.patch)
  src=$tmp/varnet_amd/csrc/vn_fused16.hip
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -I. $extra -c $src -o $tmp/vn_fused16.o
objs=$(ls *.o | grep -v "^vn_fused16.o$\|^vn_api_x.o$\|^vn_fused.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $tmp/vn_fused16.o -o ../libvarnet_hip_$name.so
rm -rf $tmp
echo built libvarnet_hip_$name.so
