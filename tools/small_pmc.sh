#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes over one small-step workload of tools/step_timeline.py (default: mor) ->
# gpurun_out/r3_small_pmc_<wl>.txt  (per kernel: mean of each counter over the dispatches after warm-up)
wl=${1:-mor}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $root
out=gpurun_out/small_pmc_$wl
rm -rf $out; mkdir -p $out
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SALU"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/$t -o p --output-format csv -- python3 tools/step_timeline.py run $wl 60 > $out/$t.log 2>&1
done
python3 - $out $wl <<'PY' > gpurun_out/r3_small_pmc_$wl.txt
import csv, glob, sys, collections
out, wl = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
print('# workload', wl, ': mean counter value per dispatch (last 40 dispatches of each kernel)')
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[-40:]
        print('   %-28s %14.0f' % (c, sum(v) / len(v)))
PY
cat gpurun_out/r3_small_pmc_$wl.txt
