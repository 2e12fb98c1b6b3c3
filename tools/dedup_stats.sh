cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ds
rocprofv3 --kernel-trace --stats -d gpurun_out/ds -o s --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
f=$(find gpurun_out/ds -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]: print('%-100s calls %5s total %9.2f ms avg %9.1f us'%(r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
find gpurun_out/ds -name "*.csv" ! -name "*kernel_stats.csv" -delete
