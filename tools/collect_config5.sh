#!/bin/bash
# rocprofv3 summaries of BASELINE config 5 (16-qubit cross_mesh) at 16 tiles of 64 points per launch:
#   bash tools/collect_config5.sh <round tag> [env assignment, e.g. QC_H2S_RB=3] [file suffix]
TAG=${1:-r03}
ENVS=${2:-QC_X=0}
SUF=${3:-}
OUT=gpurun_out/prof_${TAG}_c5$SUF
mkdir -p $OUT
export TMPDIR=/tmp
( export $ENVS; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python3 tools/bench_configs.py --skip3 --b5 1024 --steps5 3 > $OUT/c5_stats.log 2>&1 )
cp $(find $OUT/c5_stats -name "*kernel_stats.csv") $OUT/${TAG}_config5${SUF}_b1024_kernel_stats.csv
bash tools/pmc_passes.sh $OUT/c5_pmc "$ENVS" -- python3 tools/bench_configs.py --skip3 --b5 1024 --steps5 1 > /dev/null
python tools/pmc_summary.py sq $(find $OUT/c5_pmc/pass1 $OUT/c5_pmc/pass2 -name "*counter_collection.csv") > $OUT/${TAG}_config5${SUF}_b1024_sq_counters_per_kernel.csv
python tools/pmc_summary.py hbm $(find $OUT/c5_pmc/pass3 $OUT/c5_pmc/pass4 -name "*counter_collection.csv") > $OUT/${TAG}_config5${SUF}_b1024_hbm_traffic_per_kernel.csv
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $OUT
