#!/bin/bash
# config-3 part of tools/collect_profiles.sh alone:  bash tools/collect_config3.sh r02
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -- python3 tools/bench_configs.py --skip5 --steps3 3 > $OUT/c3_stats.log 2>&1
cp $(find $OUT/c3_stats -name "*kernel_stats.csv") $OUT/${TAG}_config3_b131072_kernel_stats.csv
bash tools/pmc_passes.sh $OUT/c3_pmc "QC_X=0" -- python3 tools/bench_configs.py --skip5 --b3 32768 --steps3 2 > /dev/null
python tools/pmc_summary.py sq $(find $OUT/c3_pmc/pass1 $OUT/c3_pmc/pass2 -name "*counter_collection.csv") > $OUT/${TAG}_config3_b32768_sq_counters_per_kernel.csv
python tools/pmc_summary.py hbm $(find $OUT/c3_pmc/pass3 $OUT/c3_pmc/pass4 -name "*counter_collection.csv") > $OUT/${TAG}_config3_b32768_hbm_traffic_per_kernel.csv
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls $OUT
