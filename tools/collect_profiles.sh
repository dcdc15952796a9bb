#!/bin/bash
# One GPU-box call that produces every rocprofv3 summary committed under profiles/ for a round.
#   bash tools/collect_profiles.sh <round tag, e.g. r02>
# Programs are started directly after "--" (no wrappers); counter passes are separate runs (tools/pmc_passes.sh).
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs"
# ---- headline config (bench.py): kernel stats + step timeline + counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- $BENCH > $OUT/bench_stats.log 2>&1
cp $(find $OUT/bench_stats -name "*kernel_stats.csv") $OUT/${TAG}_bench_kernel_stats.csv
python tools/trace_timeline.py $(find $OUT/bench_stats -name "*kernel_trace.csv") > $OUT/${TAG}_step_timeline.txt 2>/dev/null
bash tools/pmc_passes.sh $OUT/bench_pmc "QC_X=0" -- $BENCH > /dev/null
python tools/pmc_summary.py sq $(find $OUT/bench_pmc/pass1 $OUT/bench_pmc/pass2 -name "*counter_collection.csv") > $OUT/${TAG}_bench_sq_counters_per_kernel.csv
python tools/pmc_summary.py hbm $(find $OUT/bench_pmc/pass3 $OUT/bench_pmc/pass4 -name "*counter_collection.csv") > $OUT/${TAG}_bench_hbm_traffic_per_kernel.csv
python tools/pmc_summary.py json $(find $OUT/bench_pmc/pass3 $OUT/bench_pmc/pass4 -name "*counter_collection.csv") > $OUT/${TAG}_traffic.json
echo "bench profiles done"
# ---- config 3 (8-qubit layered x2, B = 131072)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -- python3 tools/bench_configs.py --skip5 --steps3 3 > $OUT/c3_stats.log 2>&1
cp $(find $OUT/c3_stats -name "*kernel_stats.csv") $OUT/${TAG}_config3_b131072_kernel_stats.csv
bash tools/pmc_passes.sh $OUT/c3_pmc "QC_X=0" -- python3 tools/bench_configs.py --skip5 --b3 32768 --steps3 2 > /dev/null
python tools/pmc_summary.py sq $(find $OUT/c3_pmc/pass1 $OUT/c3_pmc/pass2 -name "*counter_collection.csv") > $OUT/${TAG}_config3_b32768_sq_counters_per_kernel.csv
python tools/pmc_summary.py hbm $(find $OUT/c3_pmc/pass3 $OUT/c3_pmc/pass4 -name "*counter_collection.csv") > $OUT/${TAG}_config3_b32768_hbm_traffic_per_kernel.csv
echo "config 3 profiles done"
# ---- config 5 (16-qubit cross_mesh): 16 tiles of 64 points per launch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python3 tools/bench_configs.py --skip3 --b5 1024 --steps5 3 > $OUT/c5_stats.log 2>&1
cp $(find $OUT/c5_stats -name "*kernel_stats.csv") $OUT/${TAG}_config5_b1024_kernel_stats.csv
bash tools/pmc_passes.sh $OUT/c5_pmc "QC_X=0" -- python3 tools/bench_configs.py --skip3 --b5 1024 --steps5 1 > /dev/null
python tools/pmc_summary.py sq $(find $OUT/c5_pmc/pass1 $OUT/c5_pmc/pass2 -name "*counter_collection.csv") > $OUT/${TAG}_config5_b1024_sq_counters_per_kernel.csv
python tools/pmc_summary.py hbm $(find $OUT/c5_pmc/pass3 $OUT/c5_pmc/pass4 -name "*counter_collection.csv") > $OUT/${TAG}_config5_b1024_hbm_traffic_per_kernel.csv
echo "config 5 profiles done"
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $OUT
