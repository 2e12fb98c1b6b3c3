cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_split; rm -rf $out; mkdir -p $out
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/$t -o p --output-format csv -- python3 tools/residual_perf.py > $out/$t.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_split/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        for k in ('vn_split16_kernel<5, 13, false, 1>','vn_split16_kernel<5, 13, false, 3>','vn_pgrad16_kernel','vn_taylor16_kernel'):
            if k in n: per[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in per.items():
    med={c:sorted(v)[len(v)//2] for c,v in d.items()}
    cyc=med.get('GRBM_GUI_ACTIVE',0)/8.0
    print(k, {c:'%.3g'%v for c,v in med.items()})
    if cyc: print('   kernel cycles %.3g; MFMA pipe busy %.3f; LDS active %.3f (conflict share %.3f); wait_inst share %.3f wait_any %.3f active %.3f; VALU insts/SIMD %.3g MFMA/SIMD %.3g' % (cyc, med.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc, med.get('SQ_LDS_IDX_ACTIVE',0)/256/cyc, med.get('SQ_LDS_BANK_CONFLICT',0)/max(med.get('SQ_LDS_IDX_ACTIVE',1),1), med.get('SQ_WAIT_INST_ANY',0)/max(med.get('SQ_WAVE_CYCLES',1),1), med.get('SQ_WAIT_ANY',0)/max(med.get('SQ_WAVE_CYCLES',1),1), med.get('SQ_ACTIVE_INST_ANY',0)/max(med.get('SQ_WAVE_CYCLES',1),1), (med.get('SQ_INSTS_VALU',0)-med.get('SQ_INSTS_MFMA',0))/1024, med.get('SQ_INSTS_MFMA',0)/1024))
PY
