#!/bin/bash
# Run ON THE GPU BOX: device-side timelines of the small-step workloads (tools/step_timeline.py) -> gpurun_out/r3_timeline_<tag>.txt
tag=${1:-base}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $root
out=gpurun_out/r3_timeline_$tag.txt
: > $out
for wl in empty550 cfg2 mor cfg1 shard8; do
  rm -rf gpurun_out/tl_$wl
  rocprofv3 --kernel-trace -d gpurun_out/tl_$wl -o t --output-format csv -- python3 tools/step_timeline.py run $wl 200 >> $out 2>/dev/null
  python3 tools/step_timeline.py report gpurun_out/tl_$wl 40 >> $out 2>&1
  find gpurun_out/tl_$wl -name '*.csv' ! -name '*kernel_trace.csv' -delete
done
cat $out
