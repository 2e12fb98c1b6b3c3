#!/bin/bash
# Run ON THE GPU BOX: where the time of the reference's demos goes, kernel by kernel (rocprofv3 --kernel-trace --stats over
# examples/*.py): finds slow paths OUTSIDE the training step (monitors, re-sampling, host copies).
#   bash tools/prof_examples.sh                -> the three demos      bash tools/prof_examples.sh 2dt -> only operator_2dt (dedup)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  rm -rf gpurun_out/exprof
  S=$(date +%s.%N)
  rocprofv3 --kernel-trace --stats -d gpurun_out/exprof -o s --output-format csv -- python3 examples/$1.py $2 $3 $4 > gpurun_out/exprof_$1.log 2>&1
  E=$(date +%s.%N)
  echo "== $1 $3 $4: wall $(python3 -c "print(round($E-$S,1))") s"; grep -v rocprofv3 gpurun_out/exprof_$1.log | tail -1
  f=$(find gpurun_out/exprof -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('  total kernel time %.2f s'%(tot/1e9))
for r in rows[:9]: print('  %-70s calls %7s total %8.1f ms (%4.1f %%) avg %8.1f us'%(r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot, float(r['AverageNs'])/1e3))
PY
  rm -rf gpurun_out/exprof $2
}
if [ "$1" = "2dt" ]; then run operator_2dt gpurun_out/ex3 3000; exit 0; fi
run operator_1dt gpurun_out/ex1 30000
run operator_1dtmor gpurun_out/ex2 300
run operator_2dt gpurun_out/ex3 3000
