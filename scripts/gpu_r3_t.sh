#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3t
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 600 python -m pytest tests/test_gpu_evalfuse.py tests/test_gpu_c64.py -q --maxfail 8 > $OUT/new_tests.log 2>&1
rc=$?; echo "new tests rc=$rc" | tee -a $OUT/summary.txt; tail -12 $OUT/new_tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit 1; fi
DETAIL=1 timeout -k 10 400 python3 scripts/eval_bench.py > $OUT/eval_detail.txt 2>&1; echo "eval rc=$?" | tee -a $OUT/summary.txt
grep -v amdgpu $OUT/eval_detail.txt | head -8
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail 5 > $OUT/gpu_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc" | tee -a $OUT/summary.txt; tail -8 $OUT/gpu_tests.log | cut -c1-300
