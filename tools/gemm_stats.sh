cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gs_mine gpurun_out/gs_blas
rocprofv3 --kernel-trace --stats -d gpurun_out/gs_mine -o s --output-format csv -- python3 tools/layered_perf.py "${NET:-300,300,300}" > /dev/null 2>&1
export VN_LAYERED_ROCBLAS=1
rocprofv3 --kernel-trace --stats -d gpurun_out/gs_blas -o s --output-format csv -- python3 tools/layered_perf.py "${NET:-300,300,300}" > /dev/null 2>&1
for d in mine blas; do echo "== $d"; f=$(find gpurun_out/gs_$d -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:12]: print('%-90s calls %5s total %9.2f ms avg %9.1f us'%(r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
done
find gpurun_out/gs_mine gpurun_out/gs_blas -name "*.csv" ! -name "*kernel_stats.csv" -delete
