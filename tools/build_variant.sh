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
# the hot kernel's diagnostic blocks (-DVN_STAMPS, -DVN_FIXSTAMPS, -DVN_ABL_*) live in tools/diag/vn_fused16_diag.patch since round 6
if echo "$extra" | grep -q "VN_STAMPS\|VN_FIXSTAMPS\|VN_ABL_"; then
  if grep -q "VN_STAMPS" $tmp/t/varnet_amd/csrc/vn_fused16.hip; then :; else (cd $tmp/t && patch -s -p1 < $root/tools/diag/vn_fused16_diag.patch); fi
fi
cd $tmp/t/varnet_amd/csrc
make -j6 EXTRA="$extra" LIB=$root/varnet_amd/libvarnet_hip_$name.so > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
rm -rf $tmp
echo built $root/varnet_amd/libvarnet_hip_$name.so
