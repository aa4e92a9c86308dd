#!/bin/bash
# One GPU-box call: the whole -m gpu suite in ONE process, then a short bench line.  Usage: gpurun -- bash scripts/gpu_tests_and_bench.sh <tag> [pytest args]
set -o pipefail
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 ${PYTEST_TIMEOUT:-900} python3 -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee $OUT/summary.txt
tail -5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
if [ -z "$SKIP_BENCH" ]; then
  timeout -k 10 500 python3 bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:---no-cpu-baseline} > $OUT/bench.json 2> $OUT/bench.err; rc=$?
  echo "bench rc=$rc" | tee -a $OUT/summary.txt
  tail -3 $OUT/bench.err; cut -c1-600 $OUT/bench.json
fi
exit $rc
