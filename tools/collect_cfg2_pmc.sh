#!/bin/bash
# Run ON THE GPU BOX: only the config-2 counter passes of tools/collect_profiles.sh (a kernel edit needs the full script).
tag=${1:-r5}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $root
out=gpurun_out/prof_$tag
mkdir -p $out
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/c2pmc_$t -o p --output-format csv -- python3 bench.py --config 2 --steps 50 --warmup 5 --no-cpu-baseline --no-dedup > $out/c2pmc_$t.log 2>&1
done
find $out -name "*.csv" | wc -l
