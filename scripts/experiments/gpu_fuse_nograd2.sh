#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3ak
mkdir -p $OUT
export MAAI_WGRAD_TUNE_FILE=$PWD/$OUT/wgrad_tune.json
timeout -k 10 900 python -m pytest tests/test_gpu_evalfuse.py tests/test_gpu_model.py tests/test_driver_flow.py tests/test_gpu_dist.py -q --maxfail 5 > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $OUT/summary.txt; tail -8 $OUT/tests.log | cut -c1-250
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python3 bench.py --batch 512 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_b512.json 2> $OUT/bench_b512.err; echo "bench512 rc=$?" | tee -a $OUT/summary.txt
python3 -c "
import json
d=json.load(open('$OUT/bench_b512.json')); print('b512', d['value'], d['ms_per_step'])
" | tee -a $OUT/summary.txt
