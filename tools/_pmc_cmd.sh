cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_icache -o ic --output-format csv -- python tools/ab_perf.py base work 3 1 > gpurun_out/pmc_icache.log 2>&1
ls gpurun_out/pmc_icache
