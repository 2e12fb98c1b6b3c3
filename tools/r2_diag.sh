#!/bin/bash
# Run ON THE GPU BOX: per-phase cycle tables (VN_STAMPS builds) and the LDS bank-conflict ablation.
#   gpurun -- 'bash tools/r2_diag.sh'
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $root
out=gpurun_out/diag
mkdir -p $out
for v in stamps1w0:1 stamps1w4:1 stamps2w0:2 stamps2w3:2 stamps3w0:3; do
  n=${v%%:*}; m=${v#*:}
  echo "== $n (VN_STAMPS=$m)" >> $out/stamps.txt
  VN_STAMPS_LIB=libvarnet_hip_$n.so python tools/stamps.py 5 50 0 $m >> $out/stamps.txt 2>&1
done
python tools/ab_perf.py base,ablf,ablb,ablfb 0 5 > $out/abl_time.txt 2>&1
for n in base ablf ablb ablfb; do
  VARNET_HIP_LIB=$root/varnet_amd/libvarnet_hip_$n.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace -d $out/pmc_$n -o p --output-format csv -- python tools/quick_perf.py > $out/pmc_$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for n in ['base', 'ablf', 'ablb', 'ablfb']:
    d = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/diag/pmc_%s/**/p_counter_collection.csv' % n, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'vn_fused16_kernel' in r['Kernel_Name']:
                d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(n, {k: sorted(v)[len(v) // 2] for k, v in d.items()})
PY
