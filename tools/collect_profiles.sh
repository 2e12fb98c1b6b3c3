#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh <tag>'): kernel-trace stats and the
# separate --pmc passes the roofline numbers in bench.py / DESIGN.md come from.  Outputs land in
# gpurun_out/prof_<tag>/ ; tools/summarise_profiles.py copies the summaries into profiles/.
# (--pmc runs carry --kernel-trace only: no sys/hip/hsa trace domains beside counters.)
# The whole script no longer fits one gpurun call (1200 s): run it as two calls, `collect_profiles.sh <tag> A` (headline, stats,
# config-3 and de-duplication counters) and `collect_profiles.sh <tag> B` (configs 2, 1, 5, the other routes); each call builds
# the library from the tree on its box and takes the source hashes before its first pass.
part=${2:-all}
tag=${1:-r5}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $root
out=gpurun_out/prof_$tag
mkdir -p $out
# the library that is profiled is built HERE from the tree's sources (no stale object travels into a counter file), and the
# hash of what determines each profiled kernel's code is taken NOW: tools/summarise_profiles.py refuses a tree that has moved since
make -C varnet_amd/csrc clean > /dev/null && make -C varnet_amd/csrc -j16 > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
python3 - > $out/kernel_source_sha256_$part.json <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
print(json.dumps({k: bench.kernel_source_hash(k) for k in bench.KERNEL_SOURCES}))
PY
if [ "$part" = "D" ]; then      # only the de-duplicated formulation's passes (after an edit of vn_pgrad16.hip / vn_dedup.hip alone)
rm -rf $out/stats_dedup $out/ddpmc_*
rocprofv3 --kernel-trace --stats -d $out/stats_dedup -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $out/stats_dedup.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/ddpmc_$t -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/ddpmc_$t.log 2>&1
done
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_dedup_only.json 2> $out/bench_dedup_only.err
find $out -name "*.csv" | wc -l; exit 0
fi
if [ "$part" != "B" ]; then
python3 bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
python3 bench.py --config 2 --steps 400 --warmup 40 --no-dedup --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dedup --no-extra > $out/stats.log 2>&1
# the de-duplicated formulation: per-kernel times and HBM traffic of its four kernels (bench.py `dedup.roofline`)
rocprofv3 --kernel-trace --stats -d $out/stats_dedup -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $out/stats_dedup.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/ddpmc_$t -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/ddpmc_$t.log 2>&1
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/pmc_$t -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dedup --no-extra > $out/pmc_$t.log 2>&1
done
fi
if [ "$part" = "A" ]; then find $out -name "*.csv" | wc -l; exit 0; fi
# config 2 (small step): HBM traffic + matrix-pipe counters of its own launch (bench.py quotes them in extra.config2_small_step)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/c2pmc_$t -o p --output-format csv -- python3 bench.py --config 2 --steps 50 --warmup 5 --no-cpu-baseline --no-dedup > $out/c2pmc_$t.log 2>&1
done
# issue-cycle model of the small-step kernels (bench.py extra.*.roofline.issue_model): matrix-pipe cycles and vector instructions
# of config 1 (3x20, 96 k points) and of the config-5 mini-batch ([10,20,30], 96 k points); configs 2 and 3 use the passes above
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/c1pmc_$t -o p --output-format csv -- python3 bench.py --config 1 --steps 200 --warmup 20 --no-cpu-baseline --no-dedup > $out/c1pmc_$t.log 2>&1
  rocprofv3 --pmc $grp --kernel-trace -d $out/c5pmc_$t -o p --output-format csv -- python3 tools/step_timeline.py run mor 100 > $out/c5pmc_$t.log 2>&1
done
# the other routes: two-pass fused (integNum 216) and the generic kernels (width 64)
rocprofv3 --kernel-trace --stats -d $out/stats_q216 -o s --output-format csv -- python3 tools/q216_perf.py > $out/q216.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/stats_generic -o s --output-format csv -- python3 tools/width_perf.py 64 3 > $out/generic_w64.txt 2>&1
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/gpmc_$t -o p --output-format csv -- python3 tools/width_perf.py 64 3 > $out/gpmc_$t.log 2>&1
done
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $out/tpmc_$t -o p --output-format csv -- python3 tools/q216_perf.py > $out/tpmc_$t.log 2>&1
done
python tools/shard_perf.py > $out/shard_perf.txt 2>&1
python tools/mor_perf.py > $out/mor_perf.txt 2>&1
python tools/width_perf.py 60 4 > $out/width60.txt 2>&1
find $out -name "*.csv" | wc -l
