#!/bin/bash
# Run ON THE GPU BOX: counters of the tile kernels (vn_wide.hip) on a 3 x 128 net at config-3 size, one --pmc pass per group
# (kernel-trace only beside the counters).  Output: gpurun_out/wide_pmc/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $root
out=gpurun_out/wide_pmc
export VN_PERF_TILES_ONLY=1      # the GEMM leg of layered_perf.py under counters takes many minutes
rm -rf $out; mkdir -p $out
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/$t -o p --output-format csv -- python3 tools/layered_perf.py "${1:-128,128,128}" > $out/$t.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/wide_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'vn_wide_fwd' in n or 'bwd_kernel' in n:
            agg['fwd' if 'fwd' in n else 'bwd'][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/wide_pmc/summary.txt', 'w') as o:
    for k in sorted(agg):
        o.write(k + ' (largest launch = interior rows)\n')
        for c in sorted(agg[k]):
            o.write('  %-28s %.4g\n' % (c, max(agg[k][c])))
print(open('gpurun_out/wide_pmc/summary.txt').read())
PY
