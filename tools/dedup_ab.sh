#!/bin/bash
# Run ON THE GPU BOX: per-kernel time of the de-duplicated training step (bench.py's `dedup` leg), this tree against a side build
# (VARNET_HIP_LIB=varnet_amd/libvarnet_hip_<name>.so from tools/build_variant.sh) when one is named.   bash tools/dedup_ab.sh [name]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  rm -rf gpurun_out/ds
  rocprofv3 --kernel-trace --stats -d gpurun_out/ds -o s --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/ds_bench.json 2>/dev/null
  f=$(find gpurun_out/ds -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]: print('%-86s calls %5s total %9.2f ms avg %9.1f us'%(r['Name'][:86], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
js=json.loads([l for l in open('gpurun_out/ds_bench.json') if l.startswith('{')][-1])
print('bench: headline %.4f ms/step; dedup leg %.4f ms/step = %.3e points/s' % (js['ms_per_step'], js['dedup']['ms_per_step'], js['dedup']['value']))
PY
  find gpurun_out/ds -name "*.csv" ! -name "*kernel_stats.csv" -delete
}
echo "== this tree"; run
if [ -n "$1" ]; then echo "== side build $1"; export VARNET_HIP_LIB=$GRAFT_REPO_ROOT/varnet_amd/libvarnet_hip_$1.so; run; fi
