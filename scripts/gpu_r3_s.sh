#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 600 python -m pytest tests/test_gpu_evalfuse.py -q --maxfail 8 > $OUT/evalfuse_tests.log 2>&1
rc=$?; echo "evalfuse tests rc=$rc" | tee -a $OUT/summary.txt; tail -25 $OUT/evalfuse_tests.log | cut -c1-250
DETAIL=1 timeout -k 10 400 python3 scripts/eval_bench.py > $OUT/eval_detail.txt 2>&1; echo "eval rc=$?" | tee -a $OUT/summary.txt
grep -v amdgpu $OUT/eval_detail.txt | head -12
