#!/bin/bash
# One GPU-box call: the -m gpu suite, the bench line, and the rocprofv3 kernel-stats summary of the same command.
# Usage (from the repo root on the box): bash tools/gpu_tests_and_bench.sh <tag>
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
tail -c 3000 $OUT/bench.json
