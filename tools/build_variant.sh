#!/bin/bash
# Build the C-ABI library from another git revision (or the working tree with rev=WORK) into
# varnet_amd/libvarnet_hip_<name>.so for in-process A/B timing (tools/ab_perf.py).
set -e
rev=$1; name=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/t
if [ "$rev" = "WORK" ]; then
  mkdir -p $tmp/t/varnet_amd $tmp/t/include
  cp -r $root/varnet_amd/csrc $tmp/t/varnet_amd/csrc; cp $root/include/*.h $tmp/t/include/
  rm -f $tmp/t/varnet_amd/csrc/*.o
else
  git -C $root archive $rev varnet_amd/csrc include | tar -x -C $tmp/t
fi
cd $tmp/t/varnet_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $extra -c $f -o ${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 *.o -o $root/varnet_amd/libvarnet_hip_$name.so
rm -rf $tmp
echo built $root/varnet_amd/libvarnet_hip_$name.so
